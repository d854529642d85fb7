#!/bin/bash
# round-6 GPU call 44: occupancy of the fused residual unit / the 128 x 96 tile (waves_per_eu 3: 220 -> 111-123 registers, no scratch): codec times + tests
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 300 python tools/codec_time.py bf16x2 16,32 > $O/c44_codec_time_bf16x2.txt 2>&1; grep -v "amdgpu.ids" $O/c44_codec_time_bf16x2.txt | cut -c1-330
timeout 300 python tools/codec_time.py bf16 16 > $O/c44_codec_time_bf16.txt 2>&1; grep -v "amdgpu.ids" $O/c44_codec_time_bf16.txt | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_codec.py -x -q > $O/c44_tests.log 2>&1; tail -3 $O/c44_tests.log
