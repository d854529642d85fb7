#!/bin/bash
# round-4 GPU call 7: 64 lock-step lanes (four token tiles): tests, frame times, full-depth parity, then the bench line at --batch 64
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py -q -m gpu -k "two_panel or sixteen or lanes_equal" > $O/c7_batch_tests.log 2>&1; echo "batch tests rc=$?" | tee -a $O/c7_batch_tests.log; tail -3 $O/c7_batch_tests.log
timeout 300 python tools/batch_bench.py 0.6b 32,48,64 48 1 > $O/c7_batch_0p6b.txt 2>&1; tail -3 $O/c7_batch_0p6b.txt
timeout 300 python tools/batch_bench.py 1.7b 32,48,64 48 1 > $O/c7_batch_1p7b.txt 2>&1; tail -3 $O/c7_batch_1p7b.txt
timeout 1200 python -m pytest tests/test_gpu_batch_fulldepth.py -q -m gpu -s > $O/c7_batch_fulldepth.log 2>&1; echo "batch fulldepth rc=$?" | tee -a $O/c7_batch_fulldepth.log; grep -E "parity|passed|failed|^E " $O/c7_batch_fulldepth.log | cut -c1-250 | tail -12
timeout 1800 python bench.py --steps 5 --warmup 1 > $O/c7_bench.json 2> $O/c7_bench.err; echo "bench rc=$?"; tail -c 800 $O/c7_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/c7_bench.json").read().strip().splitlines()[-1])
for k in ("value","ttfa_ms_p50","decode_ms_per_frame"): print(k, d.get(k))
b=d.get("batched_decode_one_gpu",{}); print("batched", {k:b.get(k) for k in ("lanes","value","ms_per_lockstep_frame","decode_only_value","end_to_end_over_decode_only","streaming","kv_pool","lanes_8","lanes_16","lanes_32","valu_gemv","error")}); print("batched roofline", b.get("roofline"))
print("config3", d.get("config3_sharded_batched"))
m=d.get("model_1p7b",{}); print("1p7b", {k:m.get(k) for k in ("rtf","ttfa_ms_p50","rtf_bf16_codec","ttfa_ms_p50_bf16_codec","decode_ms_per_frame","error","batched_b32","batched_b64")})
c4=m.get("config4_voice_design_4k",{}); print("config4", {k:c4.get(k) for k in ("rtf","ttfa_ms_p50","rtf_bf16_codec","ttfa_ms_p50_bf16_codec","error")})
PY
