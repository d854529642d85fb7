#!/bin/bash
# round-6 GPU call 40: counters of the new GEMM kernels (separate --pmc passes, no trace domains): LDS bank conflicts, L2 hit rate, busy cycles
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
B=$GRAFT_REPO_ROOT/tools/microbench/gemm_bench
run() {  # name, counters..., then the shape via SHP
  local tag=$1; shift
  rm -rf /tmp/pmc_$tag
  timeout 200 rocprofv3 --pmc "$@" -d /tmp/pmc_$tag -o p -- $B 3 glds $SHP > /tmp/pmc_$tag.log 2>&1 || tail -3 /tmp/pmc_$tag.log
  DB=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  (echo "# rocprofv3 --pmc $* -- tools/microbench/gemm_bench 3 glds $SHP"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $DB) 
}
(for SHP in "2000 1024 2048" "52 1536 2048"; do
  export SHP
  run lds_$RANDOM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
  run l2_$RANDOM TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  run busy_$RANDOM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16
done) > $O/c40_pmc_gemm_kernels.txt 2>&1
grep -v "^#\|dbg\|conv_gemm\|^$" $O/c40_pmc_gemm_kernels.txt | cut -c1-200 | head -90
