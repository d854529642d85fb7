#!/bin/bash
# round-6 GPU call 2: in-kernel time stamps of the weight-stationary GEMMs of one layer (where do 6-9 us per launch go?)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
for B in 32 128; do for D in 2 4; do timeout 120 tools/microbench/skinny_trace $B 0 $D 20; done; done > $O/c2_skinny_trace_0p6b.txt 2>&1
for B in 128; do for D in 2; do timeout 120 tools/microbench/skinny_trace $B 1 $D 20; done; done > $O/c2_skinny_trace_1p7b.txt 2>&1
cat $O/c2_skinny_trace_0p6b.txt $O/c2_skinny_trace_1p7b.txt
timeout 600 tools/microbench/normfuse_bench 5 2>&1 | grep "depth 4 ==" > $O/c2_depth_check.txt; cat $O/c2_depth_check.txt
