#!/bin/bash
# round-6 GPU call 35: 64-row eight-wave tiles wired: whole GPU suite, codec times, shapes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 300 python tools/codec_time.py bf16x2 16 > $O/c35_codec_time_bf16x2.txt 2>&1; tail -5 $O/c35_codec_time_bf16x2.txt
timeout 300 python tools/codec_time.py bf16 16 > $O/c35_codec_time_bf16.txt 2>&1; tail -6 $O/c35_codec_time_bf16.txt
timeout 300 tools/microbench/gemm_bench 10 > $O/c35_gemm_shapes.txt 2>&1; grep -B3 "^chunk\|^prefill200\|^codec370 dec" $O/c35_gemm_shapes.txt | cut -c1-200
timeout 1800 python -m pytest tests -q -m gpu -x > $O/c35_tests.log 2>&1; tail -5 $O/c35_tests.log
