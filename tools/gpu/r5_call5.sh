#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 600 python tools/batch_ttfa_timeline.py 128 > $O/c5_ttfa_timeline_128.txt 2>&1; tail -40 $O/c5_ttfa_timeline_128.txt
