#!/usr/bin/env python3
"""Golden vectors for BASELINE configs[4] AT THE DEPTH IT IS BENCHMARKED AT: the CPU oracle's 28-layer prefill of a
4096-token prompt at the 1.7B shapes (hidden 2048, 16:8 heads of 128, intermediate 6144; 5 predictor layers), then
FRAMES greedy frames on top of the 4096-key cache (what /root/reference/faster_qwen3_tts/model.py:1328-1505 +
generate.py:107-134 run for a long VoiceDesign prompt).  fp32 and bf16.

    python oracle/make_golden_longprompt_full.py     # -> tests/golden/longprompt_full.npz (about 5 minutes on 8 cores)

Stored per dtype: the prefill's last-position logits and post-norm hidden state, codes [FRAMES, 16], and for every one of
the 16 x FRAMES decisions the oracle's top-1 logit and top-2 margin (same layout as tests/golden/fulldepth.npz, so
oracle/teacher_forced.py scores it).  Test infrastructure only (see oracle/qwen3tts_oracle.py header)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

from fq3hip.config import qwen3_tts_1p7b               # noqa: E402
from fq3hip.weights import synth_weights, synth_prompt  # noqa: E402
from oracle import qwen3tts_oracle as O                 # noqa: E402

L = 4096
FRAMES = 8
TRAILING = 32


def run_case(dtype: torch.dtype):
    cfg = qwen3_tts_1p7b()
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    tie, tam, tth, tpe, _ = synth_prompt(cfg, L, TRAILING, 0, dtype=dtype)
    orc = O.OracleTTS(cfg, W, max_seq_len=L + FRAMES + 8)
    orc.pred_sampling = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    sp = O.SamplingParams(max_new_tokens=FRAMES, **{**O.GREEDY, "min_new_tokens": FRAMES})
    t0 = time.time()
    seen = {}
    inner = orc.prefill

    def prefill(embeds, mask):                       # generate() prefills itself: keep what it saw
        r = inner(embeds, mask)
        seen["logits"], seen["hidden"], seen["n"], seen["t"] = r[0].clone(), r[1].clone(), r[3], time.time() - t0
        return r
    orc.prefill = prefill
    with torch.inference_mode():
        codes = orc.generate(tie, tam, tth, tpe, sp, record_margins=True)
    dt = time.time() - t0
    logits, hidden, t_pre = seen["logits"], seen["hidden"], seen["t"]
    assert seen["n"] == L
    assert codes.shape == (FRAMES, 16)
    return dict(logits=logits.float().view(-1).numpy(), hidden=hidden.float().view(-1).numpy(),
                codes=codes.numpy().astype(np.int32),
                t_margin=np.asarray(orc.margins, np.float32), t_top1=np.asarray(orc.top1, np.float32),
                p_margin=np.asarray(orc.pred_margins, np.float32).reshape(FRAMES, 15),
                p_top1=np.asarray(orc.pred_top1, np.float32).reshape(FRAMES, 15)), t_pre, dt


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    out = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        r, t_pre, dt = run_case(dtype)
        for k, v in r.items():
            out[f"1p7b_{tag}_{k}"] = v
        print(f"1.7B {tag}: prefill({L}) {t_pre:.1f}s, total {dt:.1f}s, |hidden| max {np.abs(r['hidden']).max():.3f}, "
              f"|logits| max {np.abs(r['logits']).max():.3f}, min talker margin {r['t_margin'].min():.4f}, "
              f"min predictor margin {r['p_margin'].min():.4f}", flush=True)
    out["meta"] = np.asarray([FRAMES, L, TRAILING], np.int32)
    path = os.path.join(ROOT, "tests", "golden", "longprompt_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
