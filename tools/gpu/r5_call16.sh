#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 400 tools/microbench/gemm_bench 200 skinny 64,128 > $O/c16_gemm_skinny_lanes.txt 2>&1; echo rc=$?; cut -c1-60,150-400 $O/c16_gemm_skinny_lanes.txt | tail -20
