#!/bin/bash
# round-6 GPU call 6: vocoder reference-prefix states -- bit-identity tests, timings single / batched
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_codec.py -x -q -m gpu > $O/c6_tests_codec.log 2>&1; echo "codec tests rc=$?"; tail -15 $O/c6_tests_codec.log
timeout 600 python tools/codec_time.py bf16x2 16 > $O/c6_codec_time_bf16x2.txt 2>&1; grep -E "prefix|BATCH" $O/c6_codec_time_bf16x2.txt
timeout 600 python tools/codec_time.py bf16 16 > $O/c6_codec_time_bf16.txt 2>&1; grep -E "prefix|BATCH" $O/c6_codec_time_bf16.txt
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_serving.py tests/test_gpu_batch.py -x -q -m gpu -k "not lanes_equal and not sixteen" > $O/c6_tests_api.log 2>&1; echo "api tests rc=$?"; tail -5 $O/c6_tests_api.log
