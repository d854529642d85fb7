#!/bin/bash
# round-3 GPU call 18: evidence for the bench default of the batched path (32 lanes): kernel trace of lock-step frames, FETCH_SIZE pass
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace -d /tmp/prof32 -o p -- python $GRAFT_REPO_ROOT/tools/batch_bench.py 0.6b 32 24 > /tmp/prof32.log 2>&1
 DB=$(find /tmp/prof32 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB > $O/trace_b32_0p6b.txt 2>&1)
head -24 $O/trace_b32_0p6b.txt | cut -c1-180
cd $GRAFT_REPO_ROOT
timeout 400 bash tools/pmc_pass.sh "batch 32" gpurun_out/r3/pmc_batch32_fetch.txt FETCH_SIZE
tail -4 gpurun_out/r3/pmc_batch32_fetch.txt | cut -c1-300
