#!/bin/bash
# round-4 GPU call 17: the full GPU suite and the full bench line on the 128-lane code
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/c17_gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c17_gpu_tests.log; tail -4 $O/c17_gpu_tests.log
timeout 1500 python bench.py --steps 5 --warmup 1 > $O/c17_bench.json 2> $O/c17_bench.err; echo "bench rc=$?"; tail -3 $O/c17_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/c17_bench.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ttfa_ms_p50","decode_ms_per_frame","ms_per_step")})
b=d.get("batched_decode_one_gpu",{})
for k in ("lanes","value","ms_per_lockstep_frame","decode_only_value","end_to_end_over_decode_only","streaming","streaming_32_lanes","lanes_32","lanes_64","roofline","kv_pool","valu_gemv","error"):
    print("  ",k, json.dumps(b.get(k))[:330])
print("config3", json.dumps(d.get("config3_sharded_batched"))[:500])
m=d.get("model_1p7b",{}); print("1p7b", {k:m.get(k) for k in ("rtf","ttfa_ms_p50","error")})
for k,v in m.items():
    if k.startswith("batched"): print("   ",k, json.dumps({kk:v.get(kk) for kk in ("ms_per_lockstep_frame","value")}))
print("config4", json.dumps(m.get("config4_voice_design_4k"))[:300])
PY
