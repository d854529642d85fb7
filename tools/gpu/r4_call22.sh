#!/bin/bash
# round-4 GPU call 22: the full bench line on the final code (after the probe / capture lock)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 235 python bench.py --steps 5 --warmup 1 > $O/c22_bench.json 2> $O/c22_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/c22_bench.json").read().strip().splitlines()[-1])
b=d.get("batched_decode_one_gpu",{})
print({k:d.get(k) for k in ("value","ttfa_ms_p50","decode_ms_per_frame")}, "batched", b.get("value"), b.get("ms_per_lockstep_frame"), "config3", (d.get("config3_sharded_batched") or {}).get("value"), "concurrent", json.dumps(d.get("concurrent_utterances_one_gpu")))
PY
