#!/bin/bash
# round-6 GPU call 23: kernel traces of the streaming chunk and of the batched 200-frame tail (bf16x2) on the eight-wave tiles
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
for W in chunk tail16; do
 timeout 300 rocprofv3 --kernel-trace -d /tmp/cc_$W -o p -- python $GRAFT_REPO_ROOT/tools/codec_chunk_trace.py bf16x2 $W 20 > /tmp/cc_$W.log 2>&1
 DB=$(find /tmp/cc_$W -name "*.db" | head -1)
 (echo "# rocprofv3 --kernel-trace -- python tools/codec_chunk_trace.py bf16x2 $W 20"; tail -1 /tmp/cc_$W.log; python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB) > $O/c23_codec_${W}_trace.txt 2>&1
 head -32 $O/c23_codec_${W}_trace.txt | cut -c1-175
done
