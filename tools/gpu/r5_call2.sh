#!/bin/bash
# round-5 GPU call 2: talker attention as one workgroup per (kv head, lane) (attn_decode_lane_kernel) + global (not flat) accesses through
# loaded pointers: lock-step frame A/B, the batch tests, full-depth parity, single-stream frame time (flat -> global in samplers / glue)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
FQ3_BENCH_SWEEP="attn_lane=0;attn_lane=1" timeout 400 python tools/batch_bench.py 0.6b 64,128 48 > $O/c2_batch_0p6b.txt 2>&1; tail -6 $O/c2_batch_0p6b.txt
FQ3_BENCH_SWEEP="attn_lane=0;attn_lane=2" timeout 400 python tools/batch_bench.py 0.6b 16,32 48 > $O/c2_batch_0p6b_small.txt 2>&1; tail -6 $O/c2_batch_0p6b_small.txt
FQ3_BENCH_SWEEP="attn_lane=0;attn_lane=1" timeout 400 python tools/batch_bench.py 1.7b 64,128 48 > $O/c2_batch_1p7b.txt 2>&1; tail -6 $O/c2_batch_1p7b.txt
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_fulldepth.py tests/test_gpu_paged_kv.py -x -q -m gpu > $O/c2_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c2_tests.log; tail -15 $O/c2_tests.log
cp gpurun_out/parity_batch_fulldepth.json $O/c2_parity_batch_fulldepth.json 2>/dev/null
timeout 600 python bench.py --steps 3 --warmup 1 --no-pmc --batch 0 --config3-utterances 0 --no-cpu-baseline --concurrent 0 --no-1p7b > $O/c2_bench_single.json 2> $O/c2_bench_single.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/c2_bench_single.json").read().strip().splitlines()[-1])
for k in ("value","ttfa_ms_p50","decode_ms_per_frame"): print(k, d.get(k))
print("parity", d.get("parity_bf16_frames"))
PY
