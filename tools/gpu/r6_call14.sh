#!/bin/bash
# round-6 GPU call 14: the panel kernels of <= 32 lanes on fragment-major weight copies: frozen parity counts + timing A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
FQ3_BENCH_SWEEP="packed_weights=0;packed_weights=1" timeout 900 python tools/batch_bench.py 0.6b 8,16,32 48 > $O/c14_batch_0p6b.txt 2>&1; grep "ms per lock" $O/c14_batch_0p6b.txt
FQ3_BENCH_SWEEP="packed_weights=0;packed_weights=1" timeout 900 python tools/batch_bench.py 1.7b 16,32 48 > $O/c14_batch_1p7b.txt 2>&1; grep "ms per lock" $O/c14_batch_1p7b.txt
timeout 2400 python -m pytest tests/test_gpu_batch_fulldepth.py tests/test_gpu_batch.py tests/test_gpu_decode.py -q -m gpu -x > $O/c14_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/c14_tests.log
