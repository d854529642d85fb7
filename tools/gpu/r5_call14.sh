#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 400 python tools/parity_probe_batch.py 0p6b 16,32 "attn_lane=2" > $O/c14_parity_probe_0p6b.txt 2>&1; grep -v amdgpu $O/c14_parity_probe_0p6b.txt | tail -6
timeout 400 python tools/parity_probe_batch.py 1p7b 16,32 "attn_lane=2;norm_skinny_above=16" > $O/c14_parity_probe_1p7b.txt 2>&1; grep -v amdgpu $O/c14_parity_probe_1p7b.txt | tail -8
