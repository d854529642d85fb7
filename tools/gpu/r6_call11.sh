#!/bin/bash
# round-6 GPU call 11: kernel trace of the packed prefill of 10 x 200 rows (the first wave's slices)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 300 python tools/prefill_small_time.py 0p6b 10 > $O/c11_prefill_small_10.txt 2>&1; grep -E "prefill" $O/c11_prefill_small_10.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/pp16 -o p -- python $GRAFT_REPO_ROOT/tools/prefill_small_time.py 0p6b 10 > /tmp/pp16.log 2>&1
DB=$(find /tmp/pp16 -name "*.db" | head -1)
(echo "# rocprofv3 --kernel-trace -- python tools/prefill_small_time.py 0p6b 10  (single 200-row prefills x 92 + packed prefills of 10 x 200 rows x 52)"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB) > $O/c11_packed_prefill_trace.txt 2>&1
head -40 $O/c11_packed_prefill_trace.txt | cut -c1-170
