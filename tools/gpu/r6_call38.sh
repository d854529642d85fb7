#!/bin/bash
# round-6 GPU call 38: SwiGLU in the ring tile's epilogue (many-row prefills): bit-identity tests, packed / long prefill times
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_prefill_skinny.py tests/test_gpu_longprompt.py tests/test_gpu_codec.py -x -q > $O/c38_tests.log 2>&1; tail -4 $O/c38_tests.log
timeout 300 python tools/prefill_small_time.py 0p6b 10 > $O/c38_prefill_small_10.txt 2>&1; grep -E "prefill" $O/c38_prefill_small_10.txt | tail -4
timeout 300 python tools/prefill_small_time.py 1p7b 10 > $O/c38_prefill_small_10_1p7b.txt 2>&1; grep -E "prefill" $O/c38_prefill_small_10_1p7b.txt | tail -4
timeout 300 python tools/long_prompt_check.py > $O/c38_long_prompt.txt 2>&1; tail -5 $O/c38_long_prompt.txt
