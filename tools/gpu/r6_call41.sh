#!/bin/bash
# round-6 GPU call 41: kernel trace of a group of 32 first chunks behind prefix states (the first wave's vocoding)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
for W in first32; do
 timeout 300 rocprofv3 --kernel-trace -d /tmp/cc_$W -o p -- python $GRAFT_REPO_ROOT/tools/codec_chunk_trace.py bf16x2 $W 20 > /tmp/cc_$W.log 2>&1
 DB=$(find /tmp/cc_$W -name "*.db" | head -1)
 (echo "# rocprofv3 --kernel-trace -- python tools/codec_chunk_trace.py bf16x2 $W 20"; tail -1 /tmp/cc_$W.log; python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB) > $O/c41_codec_${W}_trace.txt 2>&1
 head -30 $O/c41_codec_${W}_trace.txt | cut -c1-175
done
