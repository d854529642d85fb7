#!/bin/bash
# round-6 GPU call 21: the streaming chunk's few-row GEMMs on the eight-wave LDS-DMA tile
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(for shp in "52 1536 7168" "52 1536 14336" "104 1536 14336" "416 768 10752" "832 768 10752" "2080 384 5376" "200 1024 3072" "200 4096 1024"; do
  timeout 120 tools/microbench/gemm_bench 20 glds $shp | grep -v "plain order"
done) > $O/c21_glds_few_rows.txt 2>&1
cat $O/c21_glds_few_rows.txt
