#!/bin/bash
# round-6 GPU call 25: chain kernel on the single-tap few-row shapes of the codec's frame-level transformer (bf16 x 2: K doubled)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(for shp in "33 512 2048" "33 1536 1024" "33 2048 1024" "33 512 1024" "8 512 2048" "8 1536 1024" "64 1024 1024" "178 512 2048" "178 2048 1024" "52 1024 2048" "104 1024 2048"; do
  timeout 120 tools/microbench/gemm_bench 30 glds $shp | grep -v "plain order\|st, XCD\|128x128\|8 waves 4 st"
done) > $O/c25_chain_small.txt 2>&1
cat $O/c25_chain_small.txt
