#!/bin/bash
# round-4 GPU call 20 (final): bench line, full GPU suite, lock-step kernel traces at 32 / 64 / 128 lanes on the final code
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
HEAD=$(cat $GRAFT_REPO_ROOT/.head_for_profiles 2>/dev/null)
timeout 600 python bench.py --steps 5 --warmup 1 > $O/c20_bench.json 2> $O/c20_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/c20_bench.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ttfa_ms_p50","decode_ms_per_frame","ms_per_step")})
b=d.get("batched_decode_one_gpu",{})
for k in ("lanes","value","ms_per_lockstep_frame","decode_only_value","end_to_end_over_decode_only","streaming","streaming_32_lanes","lanes_32","lanes_64","roofline","kv_pool","error"):
    print("  ",k, json.dumps(b.get(k))[:330])
print("config3", json.dumps(d.get("config3_sharded_batched"))[:500])
m=d.get("model_1p7b",{}); print("1p7b", {k:m.get(k) for k in ("rtf","ttfa_ms_p50","error")})
for k,v in m.items():
    if k.startswith("batched"): print("   ",k, json.dumps({kk:v.get(kk) for kk in ("ms_per_lockstep_frame","value")}))
PY
timeout 560 python -m pytest tests -q -m gpu > $O/c20_gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c20_gpu_tests.log; tail -3 $O/c20_gpu_tests.log
cd /tmp && export TMPDIR=/tmp
for L in 128 64 32; do
 (timeout 120 rocprofv3 --kernel-trace -d /tmp/prof$L -o p -- python $GRAFT_REPO_ROOT/tools/batch_bench.py 0.6b $L 24 > /tmp/prof$L.log 2>&1
  DB=$(find /tmp/prof$L -name "*.db" | head -1); (echo "# rocprofv3 --kernel-trace -- python tools/batch_bench.py 0.6b $L 24  (round 4, source $HEAD)"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB) > $O/c20_batch${L}_kernel_trace.txt 2>&1)
 head -9 $O/c20_batch${L}_kernel_trace.txt | cut -c1-170
done
