#!/bin/bash
# round-6 GPU call 28: first-wave timeline at 128 lanes and the bench line on the eight-wave / chain kernels
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 600 python tools/batch_ttfa_timeline.py 128 > $O/c28_ttfa_timeline_128.txt 2>&1; tail -22 $O/c28_ttfa_timeline_128.txt
timeout 900 python bench.py --steps 5 --warmup 1 > $O/c28_bench.json 2> $O/c28_bench.err; echo "bench rc=$?"; tail -2 $O/c28_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/c28_bench.json").read().strip().splitlines()[-1])
b=d.get("batched_decode_one_gpu",{})
print({k:d.get(k) for k in ("value","ttfa_ms_p50","decode_ms_per_frame","rccl_ranks")}, "frac", d.get("roofline",{}).get("frac"), "traffic x", d.get("roofline",{}).get("traffic_over_algorithmic"))
print("batched", {k:b.get(k) for k in ("value","ms_per_lockstep_frame","decode_only_value","end_to_end_over_decode_only")}, "frac", b.get("roofline",{}).get("frac"), "traffic x", b.get("roofline",{}).get("traffic_over_algorithmic"))
for k in ("streaming","streaming_64_lanes","streaming_32_lanes"): print(k, {kk:(b.get(k) or {}).get(kk) for kk in ("value","ttfa_ms_first_wave_p50","ttfa_ms_first_wave_max","error")})
print("lanes", {k:b.get(k) for k in ("lanes_16","lanes_32","lanes_64")})
print("config3", (d.get("config3_sharded_batched") or {}).get("value"))
m=d.get("model_1p7b",{}); print("1p7b", {k:m.get(k) for k in ("rtf","ttfa_ms_p50","decode_ms_per_frame","error")}, {k:(m.get(k) or {}).get("ms_per_lockstep_frame") for k in ("batched_b32","batched_b64","batched_b128")}, "config4", {k:(m.get("config4_voice_design_4k") or {}).get(k) for k in ("rtf","ttfa_ms_p50")})
print("mfma", d.get("roofline_mfma"))
print("vocoder", {k:d.get(k) for k in d if "vocoder" in k or "codec" in k})
PY
