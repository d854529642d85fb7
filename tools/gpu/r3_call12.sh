#!/bin/bash
# round-3 GPU call 12: skinny GEMM after the wait-state fix
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 300 tools/microbench/gemm_bench 200 skinny > $O/skinny_bench2.txt 2>&1; echo "rc $?" >> $O/skinny_bench2.txt); cat $O/skinny_bench2.txt
(timeout 300 python tools/prefill_time.py 0p6b > $O/prefill_time2.txt 2>&1; echo "rc $?" >> $O/prefill_time2.txt); tail -6 $O/prefill_time2.txt
(timeout 300 python tools/prefill_time.py 1p7b >> $O/prefill_time2.txt 2>&1; echo "rc $?" >> $O/prefill_time2.txt); tail -6 $O/prefill_time2.txt
