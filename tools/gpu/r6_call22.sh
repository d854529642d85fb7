#!/bin/bash
# round-6 GPU call 22: eight-wave LDS-DMA tile also for the few-row GEMMs: codec tests, codec times, single-stream bench
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_prefill_skinny.py tests/test_gpu_longprompt.py tests/test_gpu_refenc.py tests/test_gpu_prompt.py tests/test_gpu_voice_prompt.py -x -q > $O/c22_tests.log 2>&1; tail -4 $O/c22_tests.log
timeout 300 python tools/codec_time.py bf16x2 16 > $O/c22_codec_time_bf16x2.txt 2>&1; tail -5 $O/c22_codec_time_bf16x2.txt
timeout 300 python tools/codec_time.py bf16 16 > $O/c22_codec_time_bf16.txt 2>&1; tail -6 $O/c22_codec_time_bf16.txt
timeout 300 python tools/prefill_small_time.py 0p6b 10 > $O/c22_prefill_small_10.txt 2>&1; grep -E "prefill" $O/c22_prefill_small_10.txt | tail -4
timeout 600 python tools/quick_bench.py 8 2 > $O/c22_quick_bench.txt 2>&1; tail -3 $O/c22_quick_bench.txt | cut -c1-1500
