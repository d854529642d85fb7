#!/bin/bash
# round-3 GPU call 15: prefill with the weight-stationary GEMMs + the vectorised row norm: timing, the whole GPU suite
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 300 python tools/prefill_time.py 0p6b > $O/prefill_time3.txt 2>&1; echo "rc $?" >> $O/prefill_time3.txt)
(timeout 300 python tools/prefill_time.py 1p7b >> $O/prefill_time3.txt 2>&1; echo "rc $?" >> $O/prefill_time3.txt); grep "prefill of\|logits" $O/prefill_time3.txt
(timeout 1800 python -m pytest tests -q -m gpu -x > $O/t15.log 2>&1; echo "rc $?" >> $O/t15.log); tail -6 $O/t15.log
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_pf200 -o pf -- python $GRAFT_REPO_ROOT/tools/prefill_time.py 0p6b trace > $O/prof_prefill200.log 2>&1; echo "rc $?" >> $O/prof_prefill200.log
 DB=$(find /tmp/prof_pf200 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB > $O/prefill200_kernel_trace.txt 2>&1)
head -16 $O/prefill200_kernel_trace.txt
