#!/bin/bash
# round-6 final GPU call: the whole GPU suite, smoke(), the full bench line, then the profile collection -- all on the same code
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6f; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/final_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/final_tests.log; tail -6 $O/final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.txt 2>&1; tail -2 $O/final_smoke.txt
timeout 700 python bench.py --steps 5 --warmup 1 > $O/final_bench.json 2> $O/final_bench.err; echo "bench rc=$?"; tail -2 $O/final_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6f/final_bench.json").read().strip().splitlines()[-1])
b=d.get("batched_decode_one_gpu",{})
print({k:d.get(k) for k in ("value","ttfa_ms_p50","decode_ms_per_frame","rccl_ranks")}, "frac", d.get("roofline",{}).get("frac"), "traffic x", d.get("roofline",{}).get("traffic_over_algorithmic"))
print("batched", {k:b.get(k) for k in ("value","ms_per_lockstep_frame","decode_only_value","end_to_end_over_decode_only")}, "frac", b.get("roofline",{}).get("frac"), "traffic x", b.get("roofline",{}).get("traffic_over_algorithmic"))
for k in ("streaming","streaming_64_lanes","streaming_32_lanes"): print(k, {kk:(b.get(k) or {}).get(kk) for kk in ("value","ttfa_ms_first_wave_p50","ttfa_ms_first_wave_max","error")})
print("lanes", {k:b.get(k) for k in ("lanes_16","lanes_32","lanes_64")})
print("config3", (d.get("config3_sharded_batched") or {}).get("value"))
m=d.get("model_1p7b",{}); print("1p7b", {k:m.get(k) for k in ("rtf","ttfa_ms_p50","decode_ms_per_frame","error")}, {k:(m.get(k) or {}).get("ms_per_lockstep_frame") for k in ("batched_b32","batched_b64","batched_b128")}, "config4", {k:(m.get("config4_voice_design_4k") or {}).get(k) for k in ("rtf","ttfa_ms_p50")})
print("cpu", d.get("cpu_baseline"))
PY
bash tools/gpu/r6_profiles.sh
