#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 300 tools/microbench/gemm_bench 20 2>&1 | grep -A4 "^chunk" > $O/c26_chain_taps.txt; cat $O/c26_chain_taps.txt | cut -c1-200
