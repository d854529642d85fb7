#!/bin/bash
# round-6 GPU call 5: the single-group form (ONE4) of the weight-stationary GEMM: bitwise check + timing, stamps, frame A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 600 tools/microbench/normfuse_bench 20 > $O/c5_normfuse.txt 2>&1; echo "normfuse rc=$?"; grep -E "single-group|two-tile|FAIL|self" $O/c5_normfuse.txt
(for P in -1 0; do timeout 120 tools/microbench/skinny_trace 128 0 $P 20 1; done) > $O/c5_skinny_trace_one4.txt 2>&1; grep -E "per layer|workgroups|first MFMAs|exit  |barrier|unit 0" $O/c5_skinny_trace_one4.txt
FQ3_BENCH_SWEEP="skinny_one4=0;skinny_one4=1" timeout 900 python tools/batch_bench.py 0.6b 64,96,128 48 > $O/c5_batch_0p6b.txt 2>&1; grep "ms per lock" $O/c5_batch_0p6b.txt
FQ3_BENCH_SWEEP="skinny_one4=0;skinny_one4=1" timeout 900 python tools/batch_bench.py 1.7b 64,128 48 > $O/c5_batch_1p7b.txt 2>&1; grep "ms per lock" $O/c5_batch_1p7b.txt
