#!/bin/bash
# round-6 GPU call 39: where a K step of the LDS-DMA tile goes (measurement builds: no MFMAs / no fragment reads / no barrier)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(for shp in "2000 1024 2048" "4096 4096 1024" "11832 768 5376"; do
  timeout 120 tools/microbench/gemm_bench 20 glds $shp | grep "dbg\|64x128 8 waves 3\|XCD order.*2 st\|gemm_launch, XCD"
done) > $O/c39_glds_dbg.txt 2>&1
cat $O/c39_glds_dbg.txt
