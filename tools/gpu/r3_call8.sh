#!/bin/bash
# round-3 GPU call 8: 32 lanes (two token-tile passes per launch): harness checks, parity tests, frame times; codec fuse modes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 300 tools/microbench/kernel_chain batch 10 > $O/kc_batch3.txt 2>&1; echo "rc $?" >> $O/kc_batch3.txt)
grep -c " ok$" $O/kc_batch3.txt; grep -v " ok$" $O/kc_batch3.txt | tail -24
(timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_fulldepth.py -q -m gpu -x -s > $O/t8.log 2>&1; echo "rc $?" >> $O/t8.log)
grep "parity\] batch" $O/t8.log | cut -c1-200; tail -4 $O/t8.log
(timeout 400 python tools/batch_bench.py 0.6b 16,24,32 48 > $O/bb3_0p6b.txt 2>&1)
(timeout 400 python tools/batch_bench.py 1.7b 16,32 48 > $O/bb3_1p7b.txt 2>&1)
cat $O/bb3_0p6b.txt $O/bb3_1p7b.txt | grep "ms per"
(timeout 200 python tools/codec_time.py > $O/codec_time5.txt 2>&1); cat $O/codec_time5.txt
