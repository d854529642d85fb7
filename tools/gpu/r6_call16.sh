#!/bin/bash
# round-6 GPU call 16: ring depth of the LDS-DMA 128 x 64 tile on the one-tile-per-CU shapes (packed prefill o_proj / down, codec dec.0)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(for shp in "2000 1024 2048" "2000 1024 3072" "1200 1024 2048" "3200 1024 3072" "1480 1536 7168" "1480 1536 14336" "2000 4096 1024" "11832 768 5376" "416 768 5376"; do
  timeout 120 tools/microbench/gemm_bench 20 glds $shp
done) > $O/c16_glds_depth.txt 2>&1
cat $O/c16_glds_depth.txt
