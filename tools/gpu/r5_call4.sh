#!/bin/bash
# round-5 GPU call 4: bf16x2 fused residual unit (parity + time), codec workspace sizing, done-guard of the single-stream loop, first-wave
# admission (TTFA distribution of simultaneous streaming requests; end-to-end throughput)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_paged_kv.py -x -q -m gpu > $O/c4_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c4_tests.log; tail -6 $O/c4_tests.log
for P in bf16x2; do timeout 300 python tools/codec_time.py $P 16 > $O/c4_codec_time_$P.txt 2>&1; tail -12 $O/c4_codec_time_$P.txt; done
timeout 900 python tools/batch_ttfa_probe.py 32,64,128 0,16,32,64 > $O/c4_ttfa_probe.txt 2>&1; grep "^{" $O/c4_ttfa_probe.txt
FQ3_E2E_FIRST_WAVE="0,32" timeout 600 python tools/batch_e2e_bench.py 0p6b 128 0 bf16x2 - 2 > $O/c4_e2e_128.txt 2>&1; tail -4 $O/c4_e2e_128.txt
