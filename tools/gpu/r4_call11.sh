#!/bin/bash
# round-4 GPU call 11: where the end-to-end batched run spends its time (kernel-trace timeline), decode stream priority A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(timeout 900 rocprofv3 --kernel-trace -d /tmp/e2e -o p -- python $GRAFT_REPO_ROOT/tools/batch_e2e_bench.py 0p6b 64 0 bf16x2 - 1 > $O/c11_e2e_traced.txt 2>&1; tail -1 $O/c11_e2e_traced.txt
 DB=$(find /tmp/e2e -name "*.db" | head -1); ls -la $DB; python $GRAFT_REPO_ROOT/tools/e2e_timeline.py $DB 400 > $O/c11_e2e_timeline_0p6b_64.txt 2>&1; head -40 $O/c11_e2e_timeline_0p6b_64.txt)
cd $GRAFT_REPO_ROOT
timeout 500 python tools/batch_e2e_bench.py 0p6b 64 0 bf16x2 - 2 -1 > $O/c11_e2e_prio_hi.txt 2>&1; tail -1 $O/c11_e2e_prio_hi.txt
timeout 500 python tools/batch_e2e_bench.py 0p6b 64 0 bf16x2 - 2 0 > $O/c11_e2e_prio_0.txt 2>&1; tail -1 $O/c11_e2e_prio_0.txt
