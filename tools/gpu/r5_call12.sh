#!/bin/bash
# round-5 GPU call 12: the predictor's two-token prefill as one pass over 2 B rows in the lock-step batch
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
FQ3_BENCH_SWEEP="pred_pair=0;pred_pair=1" timeout 400 python tools/batch_bench.py 0.6b 64,128 48 > $O/c12_batch_0p6b.txt 2>&1; tail -4 $O/c12_batch_0p6b.txt
FQ3_BENCH_SWEEP="pred_pair=0;pred_pair=1" timeout 400 python tools/batch_bench.py 1.7b 64,128 48 > $O/c12_batch_1p7b.txt 2>&1; tail -4 $O/c12_batch_1p7b.txt
timeout 900 python -m pytest tests/test_gpu_batch_fulldepth.py tests/test_gpu_batch.py -x -q -m gpu > $O/c12_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c12_tests.log; tail -6 $O/c12_tests.log
cp gpurun_out/parity_batch_fulldepth.json $O/c12_parity_batch_fulldepth.json 2>/dev/null
