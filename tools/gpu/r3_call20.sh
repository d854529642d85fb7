#!/bin/bash
# round-3 GPU call 20: the bench line of the final state
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench6.json 2> $O/bench6.err; echo "rc $?" >> $O/bench6.err)
tail -2 $O/bench6.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r3/bench6.json').read().strip().splitlines()[-1])
print(d['value'], d['ttfa_ms_p50'], d['decode_ms_per_frame'], d['roofline']['frac'], d['roofline']['traffic'])
print('config3', {k:v for k,v in d.get('config3_sharded_batched',{}).items() if k in ('value','seconds','lanes')})
m=d.get('model_1p7b',{})
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('ms','rtf','ttfa_ms_p50','ms_per_lockstep_frame','value','achieved','error')}) for k,v in m.items()})
b=d.get('batched_decode_one_gpu',{})
print({k:v for k,v in b.items() if k not in ('roofline','unit')})
print(d.get('roofline_mfma')); print(d.get('parity_bf16_frames'))
P
