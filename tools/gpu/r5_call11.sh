#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 300 python tools/batch_ttfa_timeline.py 128 > $O/c11_ttfa_timeline_128.txt 2>&1; tail -16 $O/c11_ttfa_timeline_128.txt
timeout 600 python tools/batch_ttfa_probe.py 32,64,128 0 > $O/c11_ttfa_probe.txt 2>&1; grep "^{" $O/c11_ttfa_probe.txt
