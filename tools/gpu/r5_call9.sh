#!/bin/bash
# round-5 GPU call 9: pipelined first wave (prompt builds under the packed prefills), lane attention without the masked tail steps
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_serving.py tests/test_gpu_api.py tests/test_gpu_paged_kv.py tests/test_gpu_batch_fulldepth.py -x -q -m gpu > $O/c9_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c9_tests.log; tail -8 $O/c9_tests.log
cp gpurun_out/parity_batch_fulldepth.json $O/c9_parity_batch_fulldepth.json 2>/dev/null
timeout 600 python tools/batch_ttfa_probe.py 32,64,128 0 > $O/c9_ttfa_probe.txt 2>&1; grep "^{" $O/c9_ttfa_probe.txt
timeout 300 python tools/batch_ttfa_timeline.py 128 > $O/c9_ttfa_timeline_128.txt 2>&1; tail -8 $O/c9_ttfa_timeline_128.txt
timeout 400 python tools/batch_bench.py 0.6b 64,128 48 > $O/c9_batch_0p6b.txt 2>&1; tail -2 $O/c9_batch_0p6b.txt
timeout 600 python tools/batch_e2e_bench.py 0p6b 128 0 bf16x2 - 2 > $O/c9_e2e_128.txt 2>&1; tail -2 $O/c9_e2e_128.txt
