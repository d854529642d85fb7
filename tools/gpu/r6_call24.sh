#!/bin/bash
# round-6 GPU call 24: the chain kernel (accumulators travel between waves) on the few-row long-K shapes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(for shp in "52 1536 7168" "52 1536 14336" "416 768 10752" "832 768 10752" "2080 384 5376" "200 1024 3072" "33 1024 2048"; do
  timeout 120 tools/microbench/gemm_bench 20 glds $shp | grep -v "plain order\|st, XCD"
done) > $O/c24_chain.txt 2>&1
cat $O/c24_chain.txt
timeout 300 tools/microbench/gemm_bench 10 2>&1 | grep -A4 "^chunk13\|^prefill200\|^codec370 dec" > $O/c24_chain_shapes.txt; cat $O/c24_chain_shapes.txt | cut -c1-200
