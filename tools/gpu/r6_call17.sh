#!/bin/bash
# round-6 GPU call 17: XCD-aware tile order of the (N tiles x M tiles) grids (LDS-DMA 128 x 64 tile, register-prefetch tiles) vs the plain order
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(for shp in "2000 1024 2048" "2000 1024 3072" "1200 1024 2048" "3200 1024 3072" "1480 1536 7168" "416 768 5376" "11832 768 5376"; do
  timeout 120 tools/microbench/gemm_bench 20 glds $shp
done) > $O/c17_xcd_order.txt 2>&1
cat $O/c17_xcd_order.txt
timeout 300 tools/microbench/gemm_bench 10 > $O/c17_gemm_shapes.txt 2>&1; cat $O/c17_gemm_shapes.txt | cut -c1-230
