#!/bin/bash
# round-6 GPU call 19: the LDS-DMA 128 x 64 tile by eight waves (512 threads) on the one-tile-per-CU shapes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(for shp in "2000 1024 2048" "2000 1024 3072" "1200 1024 2048" "3200 1024 3072" "1480 1536 7168" "416 768 5376" "11832 768 5376"; do
  timeout 120 tools/microbench/gemm_bench 20 glds $shp
done) > $O/c19_glds_8waves.txt 2>&1
cat $O/c19_glds_8waves.txt

