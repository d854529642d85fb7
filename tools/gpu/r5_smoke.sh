#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -2
