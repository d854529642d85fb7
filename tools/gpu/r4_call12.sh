#!/bin/bash
# round-4 GPU call 12: scheduler look-ahead (batch poll + one batch of frames queued ahead): tests, end-to-end A/B, streaming TTFA A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_serving.py tests/test_gpu_api.py tests/test_gpu_paged_kv.py -q -m gpu -x > $O/c12_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c12_tests.log; tail -5 $O/c12_tests.log
timeout 500 python tools/batch_e2e_bench.py 0p6b 64 0 bf16x2 - 2 0 1,0 > $O/c12_e2e_0p6b_64.txt 2>&1; tail -2 $O/c12_e2e_0p6b_64.txt
timeout 500 python tools/batch_e2e_bench.py 0p6b 32 0 bf16x2 - 2 0 1,0 > $O/c12_e2e_0p6b_32.txt 2>&1; tail -2 $O/c12_e2e_0p6b_32.txt
timeout 500 python tools/batch_e2e_bench.py 1p7b 64 0 bf16x2 - 2 0 1,0 > $O/c12_e2e_1p7b_64.txt 2>&1; tail -2 $O/c12_e2e_1p7b_64.txt
for LA in 1 0; do
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-1p7b --config3-utterances 0 --no-pmc --batch-lookahead $LA > $O/c12_bench_la$LA.json 2> $O/c12_bench_la$LA.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r4/c12_bench_la$LA.json").read().strip().splitlines()[-1])
b=d.get("batched_decode_one_gpu",{})
print("lookahead $LA: value", d.get("value"), "ttfa", d.get("ttfa_ms_p50"))
for k in ("value","decode_only","end_to_end_over_decode_only","streaming","streaming_32_lanes","lanes_32","kv_pool","error"):
    print("  ",k, json.dumps(b.get(k))[:400])
PY
done
