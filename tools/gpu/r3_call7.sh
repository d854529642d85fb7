#!/bin/bash
# round-3 GPU call 7: fused residual units (bit identity + timing), sampler early exit, then the whole GPU suite and the bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_codec.py -q -m gpu -x > $O/t7a.log 2>&1; echo "rc $?" >> $O/t7a.log); tail -4 $O/t7a.log
(timeout 200 python tools/codec_time.py > $O/codec_time4.txt 2>&1); cat $O/codec_time4.txt
(timeout 1800 python -m pytest tests -q -m gpu -x > $O/t7.log 2>&1; echo "rc $?" >> $O/t7.log); tail -6 $O/t7.log
(timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 > $O/bench3.json 2> $O/bench3.err; echo "rc $?" >> $O/bench3.err)
tail -2 $O/bench3.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r3/bench3.json').read().strip().splitlines()[-1])
print(d['value'], d['ttfa_ms_p50'], d['decode_ms_per_frame'], d['roofline']['frac'], d['roofline']['traffic'])
for k in ('config3_sharded_batched','roofline_mfma','parity_pcm'):
    print(k, d.get(k))
m=d.get('model_1p7b',{})
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('ms','rtf','ttfa_ms_p50','ms_per_lockstep_frame','value','achieved')}) for k,v in m.items()})
b=d.get('batched_decode_one_gpu',{})
print({k:v for k,v in b.items() if k!='roofline'})
P
