#!/bin/bash
# round-6 GPU call 3: per-CU delivery rate by access pattern / source; two-phase weight loads in the traced GEMM chain
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(timeout 200 tools/microbench/l2_rate_bench 64; timeout 200 tools/microbench/l2_rate_bench 128; timeout 200 tools/microbench/l2_rate_bench 256) > $O/c3_l2_rate.txt 2>&1; cat $O/c3_l2_rate.txt
(timeout 120 tools/microbench/skinny_trace 128 0 2 20 0; timeout 120 tools/microbench/skinny_trace 128 0 2 20 1) > $O/c3_skinny_trace_wphase.txt 2>&1; grep -E "per layer|workgroups|first MFMAs|exit" $O/c3_skinny_trace_wphase.txt
