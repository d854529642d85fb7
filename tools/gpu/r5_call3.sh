#!/bin/bash
# round-5 GPU call 3: lane attention, 16- vs 8-key steps (188 vs 122 VGPRs: 2 vs 4 workgroups per CU); per-kernel trace of the 128-lane
# frame (direct launches) for the split and the lane form; the batch tests again
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
FQ3_BENCH_SWEEP="attn_lane=0;attn_lane=1;attn_lane=1,attn_lane_keys=8" timeout 500 python tools/batch_bench.py 0.6b 32,64,128 48 > $O/c3_batch_0p6b.txt 2>&1; tail -9 $O/c3_batch_0p6b.txt
FQ3_BENCH_SWEEP="attn_lane=1;attn_lane=1,attn_lane_keys=8" timeout 400 python tools/batch_bench.py 1.7b 128 48 > $O/c3_batch_1p7b.txt 2>&1; tail -2 $O/c3_batch_1p7b.txt
cd /tmp && export TMPDIR=/tmp
for V in "attn_lane=0" "attn_lane=1" "attn_lane=1,attn_lane_keys=8"; do
  T=$(echo $V | tr ',=' '__')
  FQ3_BENCH_OPTS="$V" timeout 400 rocprofv3 --kernel-trace -d /tmp/kt_$T -o p -- python $GRAFT_REPO_ROOT/tools/batch_bench.py 0.6b 128 16 0 > /tmp/kt_$T.log 2>&1
  DB=$(find /tmp/kt_$T -name "*.db" | head -1)
  (echo "# rocprofv3 --kernel-trace -- FQ3_BENCH_OPTS=$V python tools/batch_bench.py 0.6b 128 16 0   (direct launches, round 5)"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB) > $O/c3_trace_$T.txt 2>&1
  grep -E "attn_decode|combine_batch|total kernel" $O/c3_trace_$T.txt | cut -c1-170
done
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_fulldepth.py tests/test_gpu_paged_kv.py -x -q -m gpu > $O/c3_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c3_tests.log; tail -8 $O/c3_tests.log
cp gpurun_out/parity_batch_fulldepth.json $O/c3_parity_batch_fulldepth.json 2>/dev/null
