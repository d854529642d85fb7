#!/bin/bash
# round-4 GPU call 13: finish events deferred behind the next frames; scheduler host-time breakdown; timeline of the new scheduler
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 500 python tools/batch_e2e_bench.py 0p6b 64 0 bf16x2 - 2 0 1 > $O/c13_e2e_0p6b_64.txt 2>&1; tail -3 $O/c13_e2e_0p6b_64.txt
timeout 500 python tools/batch_e2e_bench.py 0p6b 32 0 bf16x2 - 2 0 1 > $O/c13_e2e_0p6b_32.txt 2>&1; tail -3 $O/c13_e2e_0p6b_32.txt
timeout 500 python tools/batch_e2e_bench.py 1p7b 64 0 bf16x2 - 2 0 1 > $O/c13_e2e_1p7b_64.txt 2>&1; tail -3 $O/c13_e2e_1p7b_64.txt
cd /tmp && export TMPDIR=/tmp
(timeout 900 rocprofv3 --kernel-trace -d /tmp/e2e -o p -- python $GRAFT_REPO_ROOT/tools/batch_e2e_bench.py 0p6b 64 0 bf16x2 - 1 > $O/c13_e2e_traced.txt 2>&1; grep "real-time\|scheduler" $O/c13_e2e_traced.txt
 DB=$(find /tmp/e2e -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/e2e_timeline.py $DB 400 > $O/c13_e2e_timeline_0p6b_64.txt 2>&1; head -24 $O/c13_e2e_timeline_0p6b_64.txt)
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_batch.py tests/test_gpu_serving.py -q -m gpu -x > $O/c13_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c13_tests.log; tail -2 $O/c13_tests.log
