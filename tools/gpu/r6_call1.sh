#!/bin/bash
# round-6 GPU call 1: staging depth 4 of the weight-stationary GEMM (micro-benchmark + bitwise check), the group form of the predictor
# attention (bit-identity tests), frame A/B of both at 64 / 128 lanes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 600 tools/microbench/normfuse_bench 20 > $O/c1_normfuse.txt 2>&1; echo "normfuse rc=$?"; grep -E "depth|FAIL|self" $O/c1_normfuse.txt | grep -v "^check.*ok$" | head -80
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "pred_attention_group or lanes_equal_single_stream or lane_groups" > $O/c1_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/c1_tests.log
FQ3_BENCH_SWEEP="skinny_depth=2,pred_attn_group=0;skinny_depth=2,pred_attn_group=1;skinny_depth=4,pred_attn_group=0;skinny_depth=4,pred_attn_group=1" timeout 900 python tools/batch_bench.py 0.6b 32,64,128 48 > $O/c1_batch_0p6b.txt 2>&1; grep "ms per lock" $O/c1_batch_0p6b.txt
FQ3_BENCH_SWEEP="skinny_depth=2,pred_attn_group=0;skinny_depth=4,pred_attn_group=1" timeout 900 python tools/batch_bench.py 1.7b 64,128 48 > $O/c1_batch_1p7b.txt 2>&1; grep "ms per lock" $O/c1_batch_1p7b.txt
