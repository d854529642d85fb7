#!/bin/bash
# round-6 GPU call 9: whole-line vs half-line access patterns; the occupancy fix of the 1.7B gate|up panel kernel (parity + timing)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(timeout 200 tools/microbench/l2_rate_bench 128) > $O/c9_l2_rate.txt 2>&1; grep -v " 16 loads" $O/c9_l2_rate.txt
timeout 900 python tools/batch_bench.py 1.7b 16,32 48 > $O/c9_batch_1p7b.txt 2>&1; grep "ms per lock" $O/c9_batch_1p7b.txt
timeout 2400 python -m pytest tests/test_gpu_voice_prompt.py tests/test_gpu_batch_fulldepth.py tests/test_gpu_batch.py -q -m gpu -x > $O/c9_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/c9_tests.log
