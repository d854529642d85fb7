#!/bin/bash
# round-3 GPU call 13: PMC passes over the weight-stationary kernel (what bounds a unit: LDS, TA/TCP, L2 or issue?)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  rm -rf /tmp/pmc_$n
  timeout 120 rocprofv3 --pmc "$@" -d /tmp/pmc_$n -o p -- $GRAFT_REPO_ROOT/tools/microbench/gemm_bench 6 skinny1 416 4096 1024 1 1 > /tmp/pmc_$n.log 2>&1 || tail -3 /tmp/pmc_$n.log
  DB=$(find /tmp/pmc_$n -name "*.db" | head -1)
  (echo "# rocprofv3 --pmc $* -- gemm_bench 6 skinny1 416 4096 1024 1 1   (M=416 N=4096 K=1024, RB=1 mt=1: 256 workgroups x 26 units)"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $DB) > $O/pmc_skinny_$n.txt 2>&1
  tail -12 $O/pmc_skinny_$n.txt
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
# (TCC_* and TA_* passes time out under rocprofv3 on this stack: dropped)
