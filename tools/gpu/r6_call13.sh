#!/bin/bash
# round-6 GPU call 13: norm_fused re-measured on fragment-major weights; 17-32 lanes on the weight-stationary form (measurement only)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
FQ3_BENCH_SWEEP="norm_fused=0;norm_fused=1" timeout 900 python tools/batch_bench.py 0.6b 64,128 48 > $O/c13_batch_normfused_0p6b.txt 2>&1; grep "ms per lock" $O/c13_batch_normfused_0p6b.txt
FQ3_BENCH_SWEEP="norm_skinny_above=32;norm_skinny_above=16" timeout 900 python tools/batch_bench.py 0.6b 32 48 > $O/c13_batch_32_skinny.txt 2>&1; grep "ms per lock" $O/c13_batch_32_skinny.txt
FQ3_BENCH_SWEEP="norm_skinny_above=32;norm_skinny_above=16" timeout 900 python tools/batch_bench.py 1.7b 32 48 > $O/c13_batch_32_skinny_1p7b.txt 2>&1; grep "ms per lock" $O/c13_batch_32_skinny_1p7b.txt
