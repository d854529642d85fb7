#!/bin/bash
# round-4 GPU call 16: per-poll staging budget scaled with the lane count: end to end at 64 / 96 / 128 lanes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 600 python tools/batch_e2e_bench.py 0p6b 64 0 bf16x2 - 2 > $O/c16_e2e_0p6b_64.txt 2>&1; tail -2 $O/c16_e2e_0p6b_64.txt
timeout 600 python tools/batch_e2e_bench.py 0p6b 128 0 bf16x2 - 2 > $O/c16_e2e_0p6b_128.txt 2>&1; tail -2 $O/c16_e2e_0p6b_128.txt
timeout 600 python tools/batch_e2e_bench.py 0p6b 96 0 bf16x2 - 2 > $O/c16_e2e_0p6b_96.txt 2>&1; tail -2 $O/c16_e2e_0p6b_96.txt
timeout 600 python tools/batch_e2e_bench.py 1p7b 128 0 bf16x2 - 2 > $O/c16_e2e_1p7b_128.txt 2>&1; tail -2 $O/c16_e2e_1p7b_128.txt
timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_serving.py -q -m gpu -x > $O/c16_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c16_tests.log; tail -2 $O/c16_tests.log
