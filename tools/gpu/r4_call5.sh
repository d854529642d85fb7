#!/bin/bash
# round-4 GPU call 5: two-panel batch GEMV (chains + bit identity), batch tests, then the full bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 300 tools/microbench/kernel_chain batch > $O/c5_kernel_chain_batch.txt 2>&1; grep -E "two panels|B=32" $O/c5_kernel_chain_batch.txt | tail -14
timeout 600 python -m pytest tests/test_gpu_batch.py -q -m gpu -k "two_panel or sixteen or lanes_equal" > $O/c5_batch_tests.log 2>&1; echo "batch tests rc=$?" | tee -a $O/c5_batch_tests.log; tail -3 $O/c5_batch_tests.log
for d in 1 0; do timeout 200 python tools/batch_bench.py 0.6b 32 48 1 > $O/c5_batch32_default.txt 2>&1; done; tail -1 $O/c5_batch32_default.txt
timeout 200 python tools/batch_bench.py 1.7b 32 48 1 > $O/c5_batch32_1p7b.txt 2>&1; tail -1 $O/c5_batch32_1p7b.txt
timeout 900 python -m pytest tests/test_gpu_batch_fulldepth.py -q -m gpu > $O/c5_batch_fulldepth.log 2>&1; echo "batch fulldepth rc=$?" | tee -a $O/c5_batch_fulldepth.log; grep -E "parity|passed|failed" $O/c5_batch_fulldepth.log | tail -8
timeout 1500 python bench.py --steps 5 --warmup 1 > $O/c5_bench.json 2> $O/c5_bench.err; echo "bench rc=$?"; tail -c 1500 $O/c5_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/c5_bench.json").read().strip().splitlines()[-1])
for k in ("value","ttfa_ms_p50","decode_ms_per_frame"): print(k, d.get(k))
print("roofline", d.get("roofline"))
print("bf16 codec headline", d.get("headline_with_bf16_codec"))
b=d.get("batched_decode_one_gpu",{}); print("batched", {k:b.get(k) for k in ("value","ms_per_lockstep_frame","decode_only_value","end_to_end_over_decode_only","streaming","kv_pool","error")}); print("batched roofline", b.get("roofline"))
print("config3", d.get("config3_sharded_batched"))
print("parity_pcm", d.get("parity_pcm"))
m=d.get("model_1p7b",{}); print("1p7b", {k:m.get(k) for k in ("rtf","ttfa_ms_p50","rtf_bf16_codec","ttfa_ms_p50_bf16_codec","decode_ms_per_frame","error")}); print("config4", m.get("config4_voice_design_4k"))
print("cpu", d.get("cpu_baseline"))
PY
