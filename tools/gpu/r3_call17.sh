#!/bin/bash
# round-3 GPU call 17: rocprofv3 --kernel-trace --stats over the bench command (single-stream part), final state of the round
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-1p7b --config3-utterances 0 --batch 0 --no-pmc"
(timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_bench -o p -- $CMD > $O/bench_traced.json 2> $O/bench_traced.err; echo "rc $?" >> $O/bench_traced.err)
DB=$(find /tmp/kt_bench -name "*.db" | head -1)
(echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-1p7b --config3-utterances 0 --batch 0 --no-pmc  (round 3 final)"
 echo "# bench line of this profiled run: $(tail -1 $O/bench_traced.json | cut -c1-700)"
 python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB) > $O/bench_kernel_trace.txt 2>&1
head -30 $O/bench_kernel_trace.txt | cut -c1-200; tail -3 $O/bench_traced.err
