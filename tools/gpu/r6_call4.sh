#!/bin/bash
# round-6 GPU call 4: fragment-major weight copies -- bitwise check + per-layer timing (micro-benchmark), in-kernel stamps, product tests, frame A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 600 tools/microbench/normfuse_bench 20 > $O/c4_normfuse.txt 2>&1; echo "normfuse rc=$?"; grep -E "fragment-major|row-major|FAIL|self" $O/c4_normfuse.txt | grep -v "^check.*ok$"; grep -c "fragment-major == row-major.*ok" $O/c4_normfuse.txt
(for P in 0 1; do timeout 120 tools/microbench/skinny_trace 128 0 2 20 $P; done; for P in 0 1; do timeout 120 tools/microbench/skinny_trace 128 1 2 20 $P; done) > $O/c4_skinny_trace_packed.txt 2>&1; grep -E "per layer|workgroups|first MFMAs|exit  " $O/c4_skinny_trace_packed.txt
timeout 1500 python -m pytest tests/test_gpu_prefill_skinny.py tests/test_gpu_batch.py tests/test_gpu_decode.py -x -q -m gpu > $O/c4_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/c4_tests.log
FQ3_BENCH_SWEEP="packed_weights=0;packed_weights=1" timeout 900 python tools/batch_bench.py 0.6b 32,64,128 48 > $O/c4_batch_0p6b.txt 2>&1; grep "ms per lock" $O/c4_batch_0p6b.txt
FQ3_BENCH_SWEEP="packed_weights=0;packed_weights=1" timeout 900 python tools/batch_bench.py 1.7b 32,64,128 48 > $O/c4_batch_1p7b.txt 2>&1; grep "ms per lock" $O/c4_batch_1p7b.txt
