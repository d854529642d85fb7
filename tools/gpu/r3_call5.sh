#!/bin/bash
# round-3 GPU call 5: hardware sine in the bf16 SnakeBeta epilogues + vectorised per-column constants
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 300 tools/microbench/gemm_bench 5 codec > $O/gemm_codec2.txt 2>&1; echo "rc $?" >> $O/gemm_codec2.txt)
cat $O/gemm_codec2.txt
(timeout 200 python tools/codec_time.py > $O/codec_time2.txt 2>&1)
cat $O/codec_time2.txt
(timeout 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_api.py tests/test_gpu_serving.py -q -m gpu -x -s > $O/t5.log 2>&1; echo "rc $?" >> $O/t5.log)
grep "parity\]" $O/t5.log; tail -4 $O/t5.log
