#!/bin/bash
# round-5 GPU call 7: resident-tile flash prefill + packed attention in one launch per layer: parity, prefill times, TTFA of simultaneous requests
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_prefill_skinny.py tests/test_gpu_longprompt.py tests/test_gpu_decode.py tests/test_gpu_paged_kv.py -x -q -m gpu > $O/c7_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c7_tests.log; tail -8 $O/c7_tests.log
timeout 300 python tools/prefill_small_time.py 0p6b > $O/c7_prefill_small_0p6b.txt 2>&1; tail -10 $O/c7_prefill_small_0p6b.txt
timeout 300 python tools/prefill_small_time.py 1p7b > $O/c7_prefill_small_1p7b.txt 2>&1; tail -10 $O/c7_prefill_small_1p7b.txt
timeout 600 python tools/batch_ttfa_probe.py 32,64,128 0 > $O/c7_ttfa_probe.txt 2>&1; grep "^{" $O/c7_ttfa_probe.txt
timeout 300 python tools/batch_ttfa_timeline.py 128 > $O/c7_ttfa_timeline_128.txt 2>&1; tail -8 $O/c7_ttfa_timeline_128.txt
