#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(for shp in "2000 4096 1024" "2000 6144 1024" "3200 4096 1024" "4096 2048 2048"; do
  timeout 120 tools/microbench/gemm_bench 20 glds $shp | grep "8 waves\|gemm_launch, XCD"
done) > $O/c37_big_vs_8waves.txt 2>&1
cat $O/c37_big_vs_8waves.txt
