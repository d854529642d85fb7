#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 600 python tools/first_wave_hostprof.py 128 > $O/c32_hostprof_128.txt 2>&1; grep -v "^$" $O/c32_hostprof_128.txt | sed -n '60,140p' | cut -c1-170
