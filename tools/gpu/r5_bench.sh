#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 700 python bench.py --steps 5 --warmup 1 > $O/final2_bench.json 2> $O/final2_bench.err; echo "bench rc=$?"; tail -2 $O/final2_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/final2_bench.json").read().strip().splitlines()[-1])
b=d.get("batched_decode_one_gpu",{})
print({k:d.get(k) for k in ("value","ttfa_ms_p50","decode_ms_per_frame")}, d.get("roofline"))
print("batched", {k:b.get(k) for k in ("value","ms_per_lockstep_frame","decode_only_value","end_to_end_over_decode_only")}, b.get("roofline"))
PY
