#!/bin/bash
# round-6 GPU call 42: bf16 x 2 GEMMs on the eight-wave LDS-DMA tiles for LARGE grids too (the four-wave tile's bf16 x 2 epilogue leaves two workgroups per CU)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
for cap in 0 2048 100000; do
  timeout 300 python tools/codec_time.py bf16x2 16,32 $cap > $O/c42_codec_time_cap$cap.txt 2>&1; grep -v "fused residual units = 0\|amdgpu.ids" $O/c42_codec_time_cap$cap.txt | cut -c1-330
done
