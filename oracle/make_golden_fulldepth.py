#!/usr/bin/env python3
"""Golden ids of the CPU oracle at the BENCHMARKED configurations (full depth), for the teacher-forced GPU parity tests.

    python oracle/make_golden_fulldepth.py            # writes tests/golden/fulldepth.npz (takes a few minutes on 8 cores)

For each of {0.6B, 1.7B} x {fp32, bf16}: seeded synthetic weights at the real shapes (28 talker + 5 predictor layers),
the synthetic 200-token prompt of SURVEY.md section 8(d), FRAMES greedy frames (EOS suppressed with min_new_tokens, as
bench.py does).  Stored per case: codes [FRAMES, 16], the first token, and for every decision the oracle's top-1 logit
and its top-2 margin, so that a bf16 mismatch on the GPU can be attributed to a near-tie in units of bf16 ulps of the
logit (tests/test_gpu_fulldepth.py).  Test infrastructure only (see oracle/qwen3tts_oracle.py header).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

from fq3hip.config import qwen3_tts_0p6b, qwen3_tts_1p7b          # noqa: E402
from fq3hip.weights import synth_weights, synth_prompt            # noqa: E402
from oracle import qwen3tts_oracle as O                           # noqa: E402

FRAMES = 24
PROMPT = 200
TRAILING = 32


def run_case(size: str, dtype: torch.dtype, frames: int = FRAMES):
    cfg = qwen3_tts_0p6b() if size == "0p6b" else qwen3_tts_1p7b()
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    tie, tam, tth, tpe, _ = synth_prompt(cfg, PROMPT, TRAILING, 0, dtype=dtype)
    orc = O.OracleTTS(cfg, W, max_seq_len=PROMPT + frames + 8)
    orc.pred_sampling = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    sp = O.SamplingParams(max_new_tokens=frames, **{**O.GREEDY, "min_new_tokens": frames})
    t0 = time.time()
    with torch.inference_mode():
        codes = orc.generate(tie, tam, tth, tpe, sp, record_margins=True)
    dt = time.time() - t0
    assert codes.shape == (frames, 16)
    # decisions: talker margins has frames + 1 entries (prefill decision first), predictor frames * 15
    return dict(codes=codes.numpy().astype(np.int32),
                t_margin=np.asarray(orc.margins, np.float32), t_top1=np.asarray(orc.top1, np.float32),
                p_margin=np.asarray(orc.pred_margins, np.float32).reshape(frames, 15),
                p_top1=np.asarray(orc.pred_top1, np.float32).reshape(frames, 15)), dt


SAMPLED_FRAMES = 12
SAMPLED_SEED = 21


def sampled_noise(cfg, frames: int = SAMPLED_FRAMES, seed: int = SAMPLED_SEED):
    """Exp(1) variates shared by the oracle and the GPU run (CPU generator: identical on every machine)."""
    g = torch.Generator().manual_seed(seed)
    tn = torch.empty(frames + 1, cfg.talker.vocab_size).exponential_(1, generator=g)
    pn = torch.empty(frames, cfg.num_code_groups - 1, cfg.predictor.vocab_size).exponential_(1, generator=g)
    return tn, pn


def run_sampled_case(frames: int = SAMPLED_FRAMES):
    """0.6B fp32, product-default sampling (T 0.9, top-k 50, repetition penalty 1.05, min_new 2; predictor T 0.9 / top-k 50)
    with pre-drawn noise: the whole sampler + penalty path at the real vocabulary sizes inside the full-depth loop."""
    cfg = qwen3_tts_0p6b()
    W = synth_weights(cfg, 0, torch.float32, parts=("talker", "predictor"))
    tie, tam, tth, tpe, _ = synth_prompt(cfg, PROMPT, TRAILING, 0, dtype=torch.float32)
    tn, pn = sampled_noise(cfg, frames)
    orc = O.OracleTTS(cfg, W, max_seq_len=PROMPT + frames + 8)
    sp = O.SamplingParams(max_new_tokens=frames, min_new_tokens=frames)
    with torch.inference_mode():
        codes = orc.generate(tie, tam, tth, tpe, sp, talker_noise=tn, pred_noise=pn)
    assert codes.shape == (frames, 16)
    return codes.numpy().astype(np.int32)


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    out = {"0p6b_f32_sampled_codes": run_sampled_case()}
    print("sampled 0.6B fp32:", out["0p6b_f32_sampled_codes"][:2].tolist())
    for size in ("0p6b", "1p7b"):
        for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            r, dt = run_case(size, dtype)
            for k, v in r.items():
                out[f"{size}_{tag}_{k}"] = v
            print(f"{size} {tag}: {dt:.1f}s, min talker margin {r['t_margin'].min():.4f}, min predictor margin {r['p_margin'].min():.4f}",
                  flush=True)
    out["meta"] = np.asarray([FRAMES, PROMPT, TRAILING], np.int32)
    path = os.path.join(ROOT, "tests", "golden", "fulldepth.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
