#!/bin/bash
# round-6 GPU call 43: eight-wave tiles for every bf16 x 2 grid: codec tests, first-wave probe
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_batch.py -x -q > $O/c43_tests.log 2>&1; tail -3 $O/c43_tests.log
timeout 900 python tools/batch_ttfa_probe.py 32,64,128 0 > $O/c43_ttfa_probe.txt 2>&1; grep "^{" $O/c43_ttfa_probe.txt
timeout 600 python tools/batch_e2e_bench.py 128 > $O/c43_e2e_128.txt 2>&1; tail -4 $O/c43_e2e_128.txt | cut -c1-400
