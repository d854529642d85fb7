#!/bin/bash
# round-4 GPU call 2: paged KV -- new tests first, then the whole GPU suite, the attention chain (contiguous vs paged) and frame times
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_paged_kv.py -x -q -m gpu > $O/c2_paged.log 2>&1; echo "paged rc=$?" | tee -a $O/c2_paged.log
tail -3 $O/c2_paged.log
timeout 120 tools/microbench/kernel_chain tattn > $O/c2_kernel_chain_tattn.txt 2>&1; cat $O/c2_kernel_chain_tattn.txt | tail -6
timeout 300 python tools/quick_bench.py 0.6b 200 > $O/c2_quick_0p6b.txt 2>&1; tail -3 $O/c2_quick_0p6b.txt
timeout 300 python tools/batch_bench.py 0.6b 32 48 > $O/c2_batch32.txt 2>&1; tail -3 $O/c2_batch32.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/c2_gpu_tests.log 2>&1; echo "suite rc=$?" | tee -a $O/c2_gpu_tests.log
tail -5 $O/c2_gpu_tests.log
