#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 200 tools/microbench/kernel_chain oattn 20 > $O/c8_kernel_chain_oattn.txt 2>&1; cat $O/c8_kernel_chain_oattn.txt | tail -8
timeout 200 tools/microbench/kernel_chain pattn 20 >> $O/c8_kernel_chain_oattn.txt 2>&1; tail -3 $O/c8_kernel_chain_oattn.txt
timeout 200 tools/microbench/kernel_chain oproj 20 >> $O/c8_kernel_chain_oattn.txt 2>&1; tail -4 $O/c8_kernel_chain_oattn.txt
