#!/bin/bash
# round-3 GPU call 1: the 16-lane batch kernels (harness self-checks + chains), the new parity tests, lock-step frame times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
(timeout 300 tools/microbench/kernel_chain batch 10 > $O/kc_batch.txt 2>&1; echo "rc $?" >> $O/kc_batch.txt)
(timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_fulldepth.py tests/test_gpu_rccl.py -q -m gpu -s -x > $O/t1.log 2>&1; echo "rc $?" >> $O/t1.log)
tail -5 $O/t1.log
(timeout 400 python tools/batch_bench.py 0.6b 8,12,16 48 > $O/bb_0p6b.txt 2>&1)
(timeout 400 python tools/batch_bench.py 1.7b 8,16 48 > $O/bb_1p7b.txt 2>&1)
cat $O/bb_0p6b.txt $O/bb_1p7b.txt | grep "ms per"
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace -d /tmp/prof16 -o p -- python $GRAFT_REPO_ROOT/tools/batch_bench.py 0.6b 16 24 > /tmp/prof16.log 2>&1
 DB=$(find /tmp/prof16 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB > $GRAFT_REPO_ROOT/$O/trace_b16_0p6b.txt 2>&1)
head -30 $GRAFT_REPO_ROOT/$O/trace_b16_0p6b.txt
