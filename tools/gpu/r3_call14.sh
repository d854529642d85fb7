#!/bin/bash
# round-3 GPU call 14: weight-stationary prefill GEMMs -- the new test file, the parity suites that run a 200-token prefill, kernel trace
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_prefill_skinny.py -q -x -s > $O/t14a.log 2>&1; echo "rc $?" >> $O/t14a.log); grep "skinny prefill\|passed\|failed\|rc " $O/t14a.log | tail -14
(timeout 1200 python -m pytest tests/test_gpu_fulldepth.py tests/test_gpu_batch_fulldepth.py tests/test_gpu_decode.py tests/test_gpu_api.py tests/test_gpu_longprompt.py tests/test_gpu_batch.py tests/test_gpu_serving.py -q -x > $O/t14b.log 2>&1; echo "rc $?" >> $O/t14b.log); tail -5 $O/t14b.log
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_pf200 -o pf -- python $GRAFT_REPO_ROOT/tools/prefill_time.py 0p6b trace > $O/prof_prefill200.log 2>&1; echo "rc $?" >> $O/prof_prefill200.log
 DB=$(find /tmp/prof_pf200 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB > $O/prefill200_kernel_trace.txt 2>&1)
head -22 $O/prefill200_kernel_trace.txt
