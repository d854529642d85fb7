#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 600 python tools/ttfa_breakdown.py > $O/c45_ttfa_breakdown.txt 2>&1; tail -30 $O/c45_ttfa_breakdown.txt | cut -c1-220
