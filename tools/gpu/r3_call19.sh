#!/bin/bash
# round-3 GPU call 19: batch talker attention -- idle workers exit before loading a clamped tile: lock-step frame times, batch parity tests
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 400 python tools/batch_bench.py 0.6b 8,16,32 48 > $O/bb5_0p6b.txt 2>&1); grep "ms per" $O/bb5_0p6b.txt
(timeout 400 python tools/batch_bench.py 1.7b 32 48 > $O/bb5_1p7b.txt 2>&1); grep "ms per" $O/bb5_1p7b.txt
(timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_fulldepth.py tests/test_gpu_serving.py -q -x > $O/t19.log 2>&1; echo "rc $?" >> $O/t19.log); tail -4 $O/t19.log
