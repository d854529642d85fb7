#!/bin/bash
# round-5 GPU call 6: the whole GPU suite + the full bench line on the current code
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/c6_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c6_tests.log; tail -12 $O/c6_tests.log
timeout 600 python bench.py --steps 5 --warmup 1 > $O/c6_bench.json 2> $O/c6_bench.err; echo "bench rc=$?"; tail -3 $O/c6_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/c6_bench.json").read().strip().splitlines()[-1])
b=d.get("batched_decode_one_gpu",{})
print({k:d.get(k) for k in ("value","ttfa_ms_p50","decode_ms_per_frame","rccl_ranks")}, "roofline", d.get("roofline"))
print("batched", {k:b.get(k) for k in ("value","ms_per_lockstep_frame","decode_only_value","end_to_end_over_decode_only")}, b.get("roofline"), b.get("streaming"), b.get("streaming_32_lanes"))
print("lanes", {k:b.get(k) for k in ("lanes_16","lanes_32","lanes_64")})
print("config3", d.get("config3_sharded_batched"))
m=d.get("model_1p7b",{}); print("1p7b", {k:(v if not isinstance(v,dict) else '...') for k,v in m.items()})
print("pcm", json.dumps(d.get("parity_pcm"))[:1500])
print("mfma", d.get("roofline_mfma"))
PY
