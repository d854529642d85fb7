#!/bin/bash
# round-3 GPU call 3: codec / prefill-4k evidence (kernel trace per dispatch, MFMA counters in their own passes), the changed
# scheduler paths, in-run traffic in the bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_serving.py tests/test_gpu_api.py -q -m gpu -x > $O/t3.log 2>&1; echo "rc $?" >> $O/t3.log)
tail -4 $O/t3.log
cd /tmp && export TMPDIR=/tmp
# kernel trace (no counters): per-kernel and per-dispatch durations of two full 370-frame codec decodes
(timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_codec -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py codec > /tmp/kt_codec.log 2>&1
 DB=$(find /tmp/kt_codec -name "*.db" | head -1)
 python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB > $O/trace_codec370.txt 2>&1
 python $GRAFT_REPO_ROOT/tools/prof_timeline.py $DB 215 > $O/dispatches_codec370.txt 2>&1)
head -25 $O/trace_codec370.txt
(timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_p4k -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py prefill4k > /tmp/kt_p4k.log 2>&1
 DB=$(find /tmp/kt_p4k -name "*.db" | head -1)
 python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB > $O/trace_prefill4k.txt 2>&1)
head -12 $O/trace_prefill4k.txt
# counters, each in its own pass without any trace domain
cd $GRAFT_REPO_ROOT
timeout 400 bash tools/pmc_pass.sh codec gpurun_out/r3/pmc_codec_mfma.txt SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
timeout 400 bash tools/pmc_pass.sh prefill4k gpurun_out/r3/pmc_prefill4k_mfma.txt SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
timeout 400 bash tools/pmc_pass.sh codec gpurun_out/r3/pmc_codec_fetch.txt FETCH_SIZE
timeout 400 bash tools/pmc_pass.sh batch gpurun_out/r3/pmc_batch16_fetch.txt FETCH_SIZE
grep -E "big_gemm|glds_gemm|flash_prefill|conv_gemm" $O/pmc_codec_mfma.txt | head -20
(timeout 900 python bench.py --gpus 1 --steps 6 --warmup 2 --no-1p7b --config3-utterances 0 --concurrent 0 > $O/bench2.json 2> $O/bench2.err; echo "rc $?" >> $O/bench2.err)
tail -2 $O/bench2.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r3/bench2.json').read().strip().splitlines()[-1])
print(d['value'], d['ttfa_ms_p50'], d['roofline'])
print(d.get('batched_decode_one_gpu'))
P
