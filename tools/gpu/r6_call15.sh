#!/bin/bash
# round-6 GPU call 15: kernel trace of the 1.7B lock-step frame at 128 and 16 lanes (where does the projection / plain GEMV stand?)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
for L in 128 16; do
 timeout 400 rocprofv3 --kernel-trace -d /tmp/prof17_$L -o p -- python $GRAFT_REPO_ROOT/tools/batch_bench.py 1.7b $L 16 0 > /tmp/prof17_$L.log 2>&1
 DB=$(find /tmp/prof17_$L -name "*.db" | head -1); (echo "# rocprofv3 --kernel-trace -- python tools/batch_bench.py 1.7b $L 16 0 (direct launches; prefills included)"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB) > $O/c15_batch${L}_1p7b_trace.txt 2>&1
 head -22 $O/c15_batch${L}_1p7b_trace.txt | cut -c1-170
done
