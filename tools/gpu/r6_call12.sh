#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 300 tools/microbench/gemm_bench 10 > $O/c12_gemm_shapes.txt 2>&1; cat $O/c12_gemm_shapes.txt | cut -c1-230
