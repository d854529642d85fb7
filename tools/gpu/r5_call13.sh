#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 240 tools/microbench/normfuse_bench 20 > $O/c13_normfuse.txt 2>&1; echo "rc=$?"; grep -E "B=128|B= 32" $O/c13_normfuse.txt | grep -v check | cut -c1-150
