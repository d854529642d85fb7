#!/bin/bash
# round-3 GPU call 6: flash prefill (128-query paired blocks, packed transposed V stores), output conv rewrite
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_longprompt.py tests/test_gpu_codec.py tests/test_gpu_decode.py -q -m gpu -x -s > $O/t6.log 2>&1; echo "rc $?" >> $O/t6.log)
grep "longprompt\]" $O/t6.log; tail -5 $O/t6.log
(timeout 200 python tools/codec_time.py > $O/codec_time3.txt 2>&1); cat $O/codec_time3.txt
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_p4k -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py prefill4k > /tmp/kt_p4k.log 2>&1
 DB=$(find /tmp/kt_p4k -name "*.db" | head -1)
 python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB > $O/trace_prefill4k_2.txt 2>&1)
head -8 $O/trace_prefill4k_2.txt
(timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_codec -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py codec > /tmp/kt_codec.log 2>&1
 DB=$(find /tmp/kt_codec -name "*.db" | head -1)
 python $GRAFT_REPO_ROOT/tools/prof_timeline.py $DB 215 > $O/dispatches_codec370_2.txt 2>&1)
awk '$5+0 > 60' $O/dispatches_codec370_2.txt | head -40
