#!/bin/bash
# round-6 GPU call 20: eight-wave LDS-DMA tiles + XCD-aware tile order in the product: GEMM shapes, codec / prefill tests, packed prefill and codec times
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 300 tools/microbench/gemm_bench 10 > $O/c20_gemm_shapes.txt 2>&1; cat $O/c20_gemm_shapes.txt | cut -c1-230
timeout 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_prefill_skinny.py tests/test_gpu_longprompt.py tests/test_gpu_refenc.py tests/test_gpu_prompt.py -x -q > $O/c20_tests.log 2>&1; tail -4 $O/c20_tests.log
timeout 300 python tools/prefill_small_time.py 0p6b 10 > $O/c20_prefill_small_10.txt 2>&1; grep -E "prefill" $O/c20_prefill_small_10.txt
timeout 300 python tools/codec_time.py bf16x2 16 > $O/c20_codec_time_bf16x2.txt 2>&1; tail -6 $O/c20_codec_time_bf16x2.txt
timeout 300 python tools/codec_time.py bf16 16 > $O/c20_codec_time_bf16.txt 2>&1; tail -6 $O/c20_codec_time_bf16.txt
