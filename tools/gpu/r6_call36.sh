#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(for shp in "52 1536 14336" "52 1536 7168" "33 512 2048" "33 1536 1024" "178 512 2048" "104 1024 2048"; do
  timeout 120 tools/microbench/gemm_bench 20 glds $shp | grep "64x64 8 waves 3\|64x128 8 waves\|chain 32x32, 8 waves x 8\|gemm_launch, XCD"
done) > $O/c36_chain_vs_64rows.txt 2>&1
cat $O/c36_chain_vs_64rows.txt
