#!/bin/bash
# round-4 GPU call 4: codec -- bf16x2 mode + batched decode: tests, timings
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_codec.py -q -m gpu -s > $O/c4_codec_tests.log 2>&1; echo "codec tests rc=$?" | tee -a $O/c4_codec_tests.log
grep -E "parity|passed|failed|^E " $O/c4_codec_tests.log | tail -25
for p in bf16 bf16x2 fp32; do timeout 300 python tools/codec_time.py $p 4,16 > $O/c4_codec_time_$p.txt 2>&1; tail -4 $O/c4_codec_time_$p.txt; done
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_serving.py tests/test_gpu_api.py -q -m gpu > $O/c4_batch_tests.log 2>&1; echo "batch/serving/api tests rc=$?" | tee -a $O/c4_batch_tests.log
tail -5 $O/c4_batch_tests.log
