#!/bin/bash
# round-3 GPU call 11: weight-stationary skinny GEMM -- microbench vs the tiled kernels, the 200-token prefill on / off, kernel trace, prefill tests
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 300 tools/microbench/gemm_bench 200 skinny > $O/skinny_bench.txt 2>&1; echo "rc $?" >> $O/skinny_bench.txt); cat $O/skinny_bench.txt
(timeout 300 python tools/prefill_time.py 0p6b > $O/prefill_time.txt 2>&1; echo "rc $?" >> $O/prefill_time.txt); tail -6 $O/prefill_time.txt
(timeout 300 python tools/prefill_time.py 1p7b >> $O/prefill_time.txt 2>&1; echo "rc $?" >> $O/prefill_time.txt); tail -6 $O/prefill_time.txt
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_pf200 -o pf -- python $GRAFT_REPO_ROOT/tools/prefill_time.py 0p6b trace > $O/prof_prefill200.log 2>&1; echo "rc $?" >> $O/prof_prefill200.log
 DB=$(find /tmp/prof_pf200 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB > $O/prefill200_kernel_trace.txt 2>&1)
cd $GRAFT_REPO_ROOT
head -40 $O/prefill200_kernel_trace.txt
(timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_longprompt.py tests/test_gpu_fulldepth.py tests/test_gpu_api.py -q -x > $O/t11.log 2>&1; echo "rc $?" >> $O/t11.log); tail -5 $O/t11.log
