#!/bin/bash
# round-4 GPU call 15: 128 lock-step lanes (device-resident lane tables, rolled tile loops for 5..8 tiles)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_batch.py -q -m gpu -x -k "two_panel or sixteen_lanes or lane_groups or lanes_equal or rearm or teacher_forced or oracle_fp32" > $O/c15_tests_quick.log 2>&1; echo "quick tests rc=$?" | tee -a $O/c15_tests_quick.log; tail -3 $O/c15_tests_quick.log
timeout 600 python tools/batch_bench.py 0.6b 32,64,96,128 56 > $O/c15_frames_0p6b.txt 2>&1; grep "ms per" $O/c15_frames_0p6b.txt
timeout 600 python tools/batch_bench.py 1.7b 32,64,96,128 56 > $O/c15_frames_1p7b.txt 2>&1; grep "ms per" $O/c15_frames_1p7b.txt
timeout 600 python tools/batch_e2e_bench.py 0p6b 128 0 bf16x2 - 2 > $O/c15_e2e_0p6b_128.txt 2>&1; tail -2 $O/c15_e2e_0p6b_128.txt
timeout 600 python tools/batch_e2e_bench.py 0p6b 96 0 bf16x2 - 2 > $O/c15_e2e_0p6b_96.txt 2>&1; tail -2 $O/c15_e2e_0p6b_96.txt
timeout 900 python -m pytest tests/test_gpu_batch_fulldepth.py -q -m gpu > $O/c15_batch_fulldepth.log 2>&1; echo "fulldepth rc=$?" | tee -a $O/c15_batch_fulldepth.log; tail -3 $O/c15_batch_fulldepth.log
cp gpurun_out/parity_batch_fulldepth.json $O/c15_parity_batch_fulldepth.json 2>/dev/null
