#!/bin/bash
# round-5 GPU call 15: the vocoder stream confined to a share of the CUs (hipExtStreamCreateWithCUMask): end-to-end throughput A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
FQ3_E2E_VOC_SHARE="0,0.5,0.75,0.25" timeout 900 python tools/batch_e2e_bench.py 0p6b 128 0 bf16x2 - 2 > $O/c15_e2e_cu_share.txt 2>&1; grep -v amdgpu $O/c15_e2e_cu_share.txt | tail -14
