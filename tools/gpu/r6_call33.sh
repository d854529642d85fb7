#!/bin/bash
# round-6 GPU call 33: 64-row LDS-DMA tiles by eight waves on the streaming chunk's mid-size GEMMs
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(for shp in "416 768 10752" "832 768 10752" "2080 384 5376" "52 6144 6144" "416 3840 3072" "2080 1536 1536" "8320 192 2688" "2000 1024 2048"; do
  timeout 120 tools/microbench/gemm_bench 20 glds $shp | grep -v "plain order\|st, XCD\|8 waves 4 st\|4 waves x\|16 waves x\|8 waves x 4"
done) > $O/c33_glds_64rows.txt 2>&1
cat $O/c33_glds_64rows.txt
