#!/bin/bash
# round-4 GPU call 18: above 64 lanes the normalising GEMVs as one normalisation launch + the weight-stationary GEMM
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch_fulldepth.py -q -m gpu -s 2>&1 | grep -v Warning > $O/c18_batch_fulldepth.log; grep "parity\] batch.*B=128\|passed\|failed\|Error\|assert" $O/c18_batch_fulldepth.log | cut -c1-600
cp gpurun_out/parity_batch_fulldepth.json $O/c18_parity_batch_fulldepth.json 2>/dev/null
(FQ3_BENCH_NORM_SKINNY=0 timeout 300 python tools/batch_bench.py 0.6b 96,128 56; timeout 300 python tools/batch_bench.py 0.6b 64,96,128 56) > $O/c18_frames_0p6b.txt 2>&1; grep "ms per" $O/c18_frames_0p6b.txt
timeout 300 python tools/batch_bench.py 1.7b 96,128 56 > $O/c18_frames_1p7b.txt 2>&1; grep "ms per" $O/c18_frames_1p7b.txt
timeout 600 python tools/batch_e2e_bench.py 0p6b 128 0 bf16x2 - 2 > $O/c18_e2e_0p6b_128.txt 2>&1; tail -2 $O/c18_e2e_0p6b_128.txt
timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_serving.py -q -m gpu -x > $O/c18_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c18_tests.log; tail -2 $O/c18_tests.log
