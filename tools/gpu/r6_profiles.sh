#!/bin/bash
# round-6 profile collection (taken on the code of the commit it is run on): kernel traces of the bench command, of the 32- / 128-lane
# lock-step frame and of the codec; FETCH_SIZE of the single-stream and the batched frames (frame dispatches only)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6p; mkdir -p $O
HEAD=$(cat $GRAFT_REPO_ROOT/.head_for_profiles 2>/dev/null)
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-1p7b --config3-utterances 0 --batch 0 --no-pmc"
(timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_bench -o p -- $CMD > $O/bench_traced.json 2> $O/bench_traced.err; echo "rc $?" >> $O/bench_traced.err)
DB=$(find /tmp/kt_bench -name "*.db" | head -1)
(echo "# rocprofv3 --kernel-trace --stats -- $CMD  (round 6, source $HEAD)"
 echo "# bench line of this profiled run: $(tail -1 $O/bench_traced.json | cut -c1-900)"
 python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB) > $O/bench_kernel_trace.txt 2>&1
head -14 $O/bench_kernel_trace.txt | cut -c1-180
for L in 32 128; do
 (timeout 400 rocprofv3 --kernel-trace -d /tmp/prof$L -o p -- python $GRAFT_REPO_ROOT/tools/batch_bench.py 0.6b $L 16 0 > /tmp/prof$L.log 2>&1
  DB=$(find /tmp/prof$L -name "*.db" | head -1); (echo "# rocprofv3 --kernel-trace -- python tools/batch_bench.py 0.6b $L 16 0   (direct launches; the lanes' prefills are in the trace too; round 6, source $HEAD)"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB) > $O/batch${L}_kernel_trace.txt 2>&1)
 head -16 $O/batch${L}_kernel_trace.txt | cut -c1-180
done
for P in bf16x2; do
 (timeout 300 rocprofv3 --kernel-trace -d /tmp/profc$P -o p -- python $GRAFT_REPO_ROOT/tools/codec_time.py $P 16 > /tmp/profc$P.log 2>&1
  DB=$(find /tmp/profc$P -name "*.db" | head -1); (echo "# rocprofv3 --kernel-trace -- python tools/codec_time.py $P 16  (round 6, source $HEAD)"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB) > $O/codec_${P}_kernel_trace.txt 2>&1)
 head -12 $O/codec_${P}_kernel_trace.txt | cut -c1-180
done
cd $GRAFT_REPO_ROOT
PMC_FROM=frame_begin_kernel timeout 400 bash tools/pmc_pass.sh "frames" gpurun_out/r6p/pmc_frames_fetch.txt FETCH_SIZE; tail -3 gpurun_out/r6p/pmc_frames_fetch.txt | cut -c1-250
for L in 32 128; do PMC_FROM=frame_begin_batch_kernel timeout 400 bash tools/pmc_pass.sh "batch $L" gpurun_out/r6p/pmc_batch${L}_fetch.txt FETCH_SIZE; tail -3 gpurun_out/r6p/pmc_batch${L}_fetch.txt | cut -c1-250; done
