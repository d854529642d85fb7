#!/bin/bash
# round-4 GPU call 3: paged KV after the stream fix -- the new tests, then the whole GPU suite
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_paged_kv.py -q -m gpu > $O/c3_paged.log 2>&1; echo "paged rc=$?" | tee -a $O/c3_paged.log
tail -3 $O/c3_paged.log
timeout 1800 python -m pytest tests -q -m gpu --deselect tests/test_gpu_paged_kv.py > $O/c3_gpu_tests.log 2>&1; echo "suite rc=$?" | tee -a $O/c3_gpu_tests.log
tail -8 $O/c3_gpu_tests.log
