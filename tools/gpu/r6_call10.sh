#!/bin/bash
# round-6 GPU call 10: whole-line (PAIR) copies in the 256-wide ring tile: bitwise checks + timings, codec tests + timing
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 300 tools/microbench/gemm_bench 10 sweep > $O/c10_gemm_sweep.txt 2>&1; head -12 $O/c10_gemm_sweep.txt
timeout 300 tools/microbench/gemm_bench 10 > $O/c10_gemm_shapes.txt 2>&1; cat $O/c10_gemm_shapes.txt
timeout 1500 python -m pytest tests/test_gpu_codec.py tests/test_gpu_longprompt.py tests/test_gpu_prefill_skinny.py -x -q -m gpu > $O/c10_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/c10_tests.log
timeout 600 python tools/codec_time.py bf16x2 16 > $O/c10_codec_time_bf16x2.txt 2>&1; grep -E "precision" $O/c10_codec_time_bf16x2.txt | head -6
timeout 600 python tools/codec_time.py bf16 16 > $O/c10_codec_time_bf16.txt 2>&1; grep -E "precision" $O/c10_codec_time_bf16.txt | head -6
