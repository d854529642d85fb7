#!/bin/bash
# round-3 GPU call 4: LDS-parked GEMM epilogues -- harness (bit-identity vs the register-layout epilogue + timings), the tests of
# every path that uses the GEMM kernels, codec decode times
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 300 tools/microbench/gemm_bench 5 codec > $O/gemm_codec.txt 2>&1; echo "rc $?" >> $O/gemm_codec.txt)
cat $O/gemm_codec.txt
(timeout 200 tools/microbench/gemm_bench 10 > $O/gemm_bench.txt 2>&1; echo "rc $?" >> $O/gemm_bench.txt)
cat $O/gemm_bench.txt
(timeout 200 python tools/codec_time.py > $O/codec_time.txt 2>&1; timeout 200 python tools/codec_time.py fp32 >> $O/codec_time.txt 2>&1)
cat $O/codec_time.txt
(timeout 1200 python -m pytest tests/test_gpu_codec.py tests/test_gpu_refenc.py tests/test_gpu_prompt.py tests/test_gpu_longprompt.py tests/test_gpu_decode.py tests/test_gpu_voice_prompt.py -q -m gpu -x > $O/t4.log 2>&1; echo "rc $?" >> $O/t4.log)
tail -6 $O/t4.log
