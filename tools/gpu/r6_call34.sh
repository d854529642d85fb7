#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(for shp in "3200 1024 3072" "1480 1536 7168" "1200 1024 2048" "2000 1024 3072" "2000 2048 2048" "2000 2048 6144" "200 6144 1024" "4160 384 5376"; do
  timeout 120 tools/microbench/gemm_bench 20 glds $shp | grep -v "plain order\|st, XCD\|8 waves 4 st\|4 waves x\|16 waves x\|8 waves x\|PF=4"
done) > $O/c34_glds_64rows_b.txt 2>&1
cat $O/c34_glds_64rows_b.txt
