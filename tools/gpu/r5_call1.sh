#!/bin/bash
# round-5 GPU call 1: the RMSNorm folded into the weight-stationary GEMM pair -- kernel self-checks + per-layer chains, lock-step frame A/B,
# full-depth parity of the batch, the prefill's GEMM tests (the issue-order change touches every skinny instantiation)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 180 tools/microbench/normfuse_bench 20 > $O/c1_normfuse.txt 2>&1; echo "normfuse rc=$?"; tail -60 $O/c1_normfuse.txt
FQ3_BENCH_SWEEP="norm_fused=0;norm_fused=1" timeout 400 python tools/batch_bench.py 0.6b 64,128 48 > $O/c1_batch_0p6b.txt 2>&1; cat $O/c1_batch_0p6b.txt | tail -8
FQ3_BENCH_SWEEP="norm_fused=1;norm_fused=1,norm_skinny_above=8,skinny=2;norm_fused=0,norm_skinny_above=8,skinny=2" timeout 400 python tools/batch_bench.py 0.6b 16,32 48 > $O/c1_batch_0p6b_small.txt 2>&1; tail -8 $O/c1_batch_0p6b_small.txt
FQ3_BENCH_SWEEP="norm_fused=0;norm_fused=1" timeout 400 python tools/batch_bench.py 1.7b 64,128 48 > $O/c1_batch_1p7b.txt 2>&1; tail -6 $O/c1_batch_1p7b.txt
timeout 900 python -m pytest tests/test_gpu_batch_fulldepth.py tests/test_gpu_prefill_skinny.py -x -q -m gpu > $O/c1_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c1_tests.log; tail -15 $O/c1_tests.log
cp gpurun_out/parity_batch_fulldepth.json $O/c1_parity_batch_fulldepth.json 2>/dev/null
