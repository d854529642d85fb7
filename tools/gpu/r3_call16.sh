#!/bin/bash
# round-3 GPU call 16: the whole GPU suite, smoke(), the bench line (state: weight-stationary prefill GEMMs + vectorised row norm)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
(timeout 1800 python -m pytest tests -q -m gpu > $O/t16.log 2>&1; echo "rc $?" >> $O/t16.log); tail -8 $O/t16.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log); tail -2 $O/smoke.log
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench5.json 2> $O/bench5.err; echo "rc $?" >> $O/bench5.err)
tail -2 $O/bench5.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r3/bench5.json').read().strip().splitlines()[-1])
print(d['value'], d['ttfa_ms_p50'], d['decode_ms_per_frame'], d['roofline']['frac'], d['roofline']['traffic'])
print('config3', d.get('config3_sharded_batched'))
m=d.get('model_1p7b',{})
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('ms','rtf','ttfa_ms_p50','ms_per_lockstep_frame','value','achieved','error')}) for k,v in m.items()})
b=d.get('batched_decode_one_gpu',{})
print({k:v for k,v in b.items() if k!='roofline'})
print(d.get('roofline_mfma')); print(d.get('parity_bf16_frames'))
P
