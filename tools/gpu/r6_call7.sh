#!/bin/bash
# round-6 GPU call 7: first chunks in larger vocoder groups; first-wave TTFA probe; end-to-end batch
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 900 python tools/codec_time.py bf16x2 32,64 > $O/c7_codec_time_bf16x2_groups.txt 2>&1; grep -E "BATCH" $O/c7_codec_time_bf16x2_groups.txt
timeout 900 python tools/batch_ttfa_probe.py 32,64,128 0 > $O/c7_ttfa_probe.txt 2>&1; grep "^{" $O/c7_ttfa_probe.txt
timeout 900 python tools/batch_e2e_bench.py 0p6b 128 0 bf16x2 - 2 > $O/c7_e2e_128.txt 2>&1; tail -3 $O/c7_e2e_128.txt
