#!/bin/bash
# round-6 GPU call 29: first wave: each slice's first tokens read behind ITS prefill, lanes armed while the later slices prefill
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 600 python tools/batch_ttfa_timeline.py 128 > $O/c29_ttfa_timeline_128.txt 2>&1; tail -14 $O/c29_ttfa_timeline_128.txt
timeout 900 python tools/batch_ttfa_probe.py 32,64,128 0 > $O/c29_ttfa_probe.txt 2>&1; grep "^{" $O/c29_ttfa_probe.txt
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_serving.py tests/test_gpu_api.py -x -q > $O/c29_tests.log 2>&1; tail -3 $O/c29_tests.log
