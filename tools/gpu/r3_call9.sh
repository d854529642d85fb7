#!/bin/bash
# round-3 GPU call 9: compile-time token-tile count (straight-line 16-lane kernels again, early second-tile loads at 32 lanes)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 300 tools/microbench/kernel_chain batch 10 > $O/kc_batch4.txt 2>&1; echo "rc $?" >> $O/kc_batch4.txt)
grep -c " ok$" $O/kc_batch4.txt; grep -v " ok$" $O/kc_batch4.txt | tail -20
(timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_fulldepth.py tests/test_gpu_codec.py -q -m gpu -x > $O/t9.log 2>&1; echo "rc $?" >> $O/t9.log)
tail -4 $O/t9.log
(timeout 400 python tools/batch_bench.py 0.6b 8,16,32 48 > $O/bb4_0p6b.txt 2>&1)
(timeout 400 python tools/batch_bench.py 1.7b 16,32 48 > $O/bb4_1p7b.txt 2>&1)
cat $O/bb4_0p6b.txt $O/bb4_1p7b.txt | grep "ms per"
