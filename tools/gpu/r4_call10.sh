#!/bin/bash
# round-4 GPU call 10: lane groups (two lock-step chains on concurrent streams) + the packed normalisation prologue
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_batch.py -q -m gpu -k "lane_groups or two_panel or sixteen_lanes" > $O/c10_tests_quick.log 2>&1; echo "quick tests rc=$?" | tee -a $O/c10_tests_quick.log; tail -3 $O/c10_tests_quick.log
(timeout 500 python tools/batch_bench.py 0.6b 32,48,64 56 1 - 1,2 ; timeout 300 python tools/batch_bench.py 0.6b 48,64 56 1 - 3,4) > $O/c10_groups_0p6b.txt 2>&1; grep "ms per" $O/c10_groups_0p6b.txt
timeout 500 python tools/batch_bench.py 1.7b 32,64 56 1 - 1,2 > $O/c10_groups_1p7b.txt 2>&1; grep "ms per" $O/c10_groups_1p7b.txt
timeout 500 python tools/batch_e2e_bench.py 0p6b 64 0 bf16x2 1,2 > $O/c10_e2e_0p6b_64.txt 2>&1; tail -2 $O/c10_e2e_0p6b_64.txt
timeout 500 python tools/batch_e2e_bench.py 1p7b 64 0 bf16x2 1,2 > $O/c10_e2e_1p7b_64.txt 2>&1; tail -2 $O/c10_e2e_1p7b_64.txt
timeout 600 python -m pytest tests/test_gpu_batch_fulldepth.py -q -m gpu > $O/c10_batch_fulldepth.log 2>&1; echo "fulldepth rc=$?" | tee -a $O/c10_batch_fulldepth.log; tail -3 $O/c10_batch_fulldepth.log
cp gpurun_out/parity_batch_fulldepth.json $O/c10_parity_batch_fulldepth.json 2>/dev/null
