#!/bin/bash
# round-4 GPU call 9: probed side streams + incremental vocoding A/B, then single-stream TTFA stability
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 600 python tools/batch_e2e_bench.py 0p6b 64 0,64,100 > $O/c9_e2e_0p6b_64.txt 2>&1; tail -3 $O/c9_e2e_0p6b_64.txt
timeout 600 python tools/batch_e2e_bench.py 0p6b 32 0,64 > $O/c9_e2e_0p6b_32.txt 2>&1; tail -2 $O/c9_e2e_0p6b_32.txt
timeout 600 python tools/batch_e2e_bench.py 1p7b 64 0,64 > $O/c9_e2e_1p7b_64.txt 2>&1; tail -2 $O/c9_e2e_1p7b_64.txt
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_serving.py tests/test_gpu_api.py -q -m gpu > $O/c9_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c9_tests.log; tail -3 $O/c9_tests.log
timeout 900 python bench.py --steps 5 --warmup 1 --no-pmc --batch 0 --config3-utterances 0 --no-cpu-baseline --concurrent 0 > $O/c9_bench_single.json 2> $O/c9_bench_single.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/c9_bench_single.json").read().strip().splitlines()[-1])
for k in ("value","ttfa_ms_p50","ttfa_ms_mean","decode_ms_per_frame"): print(k, d.get(k))
print("bf16 codec headline", d.get("headline_with_bf16_codec"))
m=d.get("model_1p7b",{}); print("1p7b", {k:m.get(k) for k in ("rtf","ttfa_ms_p50","rtf_bf16_codec","ttfa_ms_p50_bf16_codec","error")})
c4=m.get("config4_voice_design_4k",{}); print("config4", {k:c4.get(k) for k in ("rtf","ttfa_ms_p50","rtf_bf16_codec","ttfa_ms_p50_bf16_codec","error")})
PY
