#!/bin/bash
# round-6 GPU call 31: chunk groups launched as they fill + host ids noted: first-wave TTFA, streaming tests
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 900 python tools/batch_ttfa_probe.py 32,64,128 0 > $O/c31_ttfa_probe.txt 2>&1; grep "^{" $O/c31_ttfa_probe.txt
timeout 600 python tools/batch_ttfa_timeline.py 128 > $O/c31_ttfa_timeline_128.txt 2>&1; tail -14 $O/c31_ttfa_timeline_128.txt
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_serving.py tests/test_gpu_api.py tests/test_gpu_voice_prompt.py tests/test_gpu_prompt.py -x -q > $O/c31_tests.log 2>&1; tail -3 $O/c31_tests.log
