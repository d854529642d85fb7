#!/bin/bash
# round-4 GPU call 14: admissions without a device wait, request source under the prefill stream: tests, e2e, the full bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_serving.py tests/test_gpu_api.py tests/test_gpu_paged_kv.py -q -m gpu -x > $O/c14_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c14_tests.log; tail -2 $O/c14_tests.log
timeout 500 python tools/batch_e2e_bench.py 0p6b 64 0 bf16x2 - 2 0 1,0 > $O/c14_e2e_0p6b_64.txt 2>&1; tail -4 $O/c14_e2e_0p6b_64.txt
timeout 500 python tools/batch_e2e_bench.py 0p6b 32 0 bf16x2 - 2 0 1 > $O/c14_e2e_0p6b_32.txt 2>&1; tail -2 $O/c14_e2e_0p6b_32.txt
timeout 500 python tools/batch_e2e_bench.py 1p7b 64 0 bf16x2 - 2 0 1 > $O/c14_e2e_1p7b_64.txt 2>&1; tail -2 $O/c14_e2e_1p7b_64.txt
timeout 1200 python bench.py --steps 5 --warmup 1 > $O/c14_bench.json 2> $O/c14_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/c14_bench.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ttfa_ms_p50","decode_ms_per_frame","ms_per_step")})
b=d.get("batched_decode_one_gpu",{})
for k in ("value","ms_per_lockstep_frame","decode_only_value","end_to_end_over_decode_only","streaming","streaming_32_lanes","lanes_32","roofline","error"):
    print("  ",k, json.dumps(b.get(k))[:300])
print("config3", json.dumps(d.get("config3_sharded_batched"))[:400])
m=d.get("model_1p7b",{}); print("1p7b", {k:m.get(k) for k in ("rtf","ttfa_ms_p50","error")}); print("  batched", json.dumps({k:v for k,v in m.items() if k.startswith("batched")})[:500])
PY
