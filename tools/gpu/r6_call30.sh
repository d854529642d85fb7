#!/bin/bash
# round-6 GPU call 30: per-group delivery of a wave's first chunks; host profile of the first wave
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 900 python tools/batch_ttfa_probe.py 32,64,128 0 > $O/c30_ttfa_probe.txt 2>&1; grep "^{" $O/c30_ttfa_probe.txt
timeout 600 python tools/first_wave_hostprof.py 128 > $O/c30_hostprof_128.txt 2>&1; grep -v "^$" $O/c30_hostprof_128.txt | head -70 | cut -c1-180
