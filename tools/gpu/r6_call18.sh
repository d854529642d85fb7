#!/bin/bash
# round-6 GPU call 18: steady-state per-CU operand delivery: registers vs registers + ds_write vs LDS-DMA
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
(timeout 120 tools/microbench/dma_rate_bench 4000 4096 1024; timeout 120 tools/microbench/dma_rate_bench 4000 4096 4096) > $O/c18_dma_rate.txt 2>&1
cat $O/c18_dma_rate.txt
