#!/bin/bash
# round-4 GPU call 21: the concurrent-streams block after serialising stream probes with graph captures
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-1p7b --config3-utterances 0 --batch 0 --no-pmc > $O/c21_bench.json 2> $O/c21_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/c21_bench.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ttfa_ms_p50")}, json.dumps(d.get("concurrent_utterances_one_gpu")))
PY
