#!/bin/bash
# round-4 GPU call 8: incremental (exact) vocoding in the batch path: test, then the bench line (64 lanes)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py -q -m gpu -k "incremental or equals_single or custom_voice_and" > $O/c8_batch_tests.log 2>&1; echo "batch tests rc=$?" | tee -a $O/c8_batch_tests.log; grep -E "passed|failed|^E " $O/c8_batch_tests.log | cut -c1-300 | tail -8
timeout 1800 python bench.py --steps 5 --warmup 1 --no-pmc > $O/c8_bench.json 2> $O/c8_bench.err; echo "bench rc=$?"; tail -c 600 $O/c8_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/c8_bench.json").read().strip().splitlines()[-1])
for k in ("value","ttfa_ms_p50","decode_ms_per_frame"): print(k, d.get(k))
b=d.get("batched_decode_one_gpu",{}); print("batched", {k:b.get(k) for k in ("lanes","value","ms_per_lockstep_frame","decode_only_value","end_to_end_over_decode_only","streaming","streaming_32_lanes","kv_pool","error")})
print("config3", d.get("config3_sharded_batched"))
m=d.get("model_1p7b",{}); print("1p7b", {k:m.get(k) for k in ("rtf","ttfa_ms_p50","rtf_bf16_codec","ttfa_ms_p50_bf16_codec","error")})
PY
