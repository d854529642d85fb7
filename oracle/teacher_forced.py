"""Teacher-forced scoring of the HIP decode loop against golden oracle ids (TEST INFRASTRUCTURE, not the product:
imported only by tests/, bench.py's parity note and __graft_entry__.smoke()).

The oracle (oracle/qwen3tts_oracle.py, run by oracle/make_golden_fulldepth.py) decoded N greedy frames on its own; the
HIP loop is armed with the same prompt and re-plays those N frames through its real fused path (hipGraph replay) with
``fq3_decode_set_forced``: every sampler records ITS OWN arg-max and continues with the oracle's id, so that every one
of the 16 x N decisions is scored under an identical history, instead of stopping at the first divergence.  This is the
decision-level form of the relation the reference's own tests pin (tests/test_e2e_parity.py:414-427: fast path ids ==
upstream ids under greedy decoding).

A mismatch is *attributed to a near-tie* when the oracle's top-2 margin at that decision is at most ``k_ulp`` bf16 ulps
of the winning logit (ulp = 2^(floor(log2|x|) - 7)); fp32 contexts must match everywhere.
"""
from __future__ import annotations

import numpy as np
import torch


def bf16_ulp(x: np.ndarray) -> np.ndarray:
    ax = np.maximum(np.abs(x.astype(np.float64)), 2.0 ** -126)
    return 2.0 ** (np.floor(np.log2(ax)) - 7)


def load_case(npz, prefix: str):
    return {k: npz[f"{prefix}_{k}"] for k in ("codes", "t_margin", "t_top1", "p_margin", "p_top1")}


def forced_decisions(eng, cfg, tie, tth, tpe, codes: np.ndarray, graph: bool = True):
    """Run prefill + len(codes) teacher-forced frames.  Returns int64 [N, 16] HIP decisions aligned with `codes`."""
    dev = eng.device
    N, G = codes.shape
    V, eos = cfg.talker.vocab_size, cfg.codec_eos_token_id
    greedy = dict(temperature=1.0, top_k=0, top_p=1.0, do_sample=False)
    logits, hidden = eng.prefill(tie[0].to(dev).contiguous())
    tok0 = eng.sample(logits, sup_lo=max(0, V - 1024), sup_hi=V, keep_id=eos, suppress_eos=True, **greedy)
    forced = torch.zeros(N + 1, G, dtype=torch.int32)
    forced[:N] = torch.from_numpy(codes.astype(np.int32))
    forced[N, 0] = int(codes[N - 1, 0])                 # continuation id after the last scored frame (never scored)
    forced = forced.to(dev).contiguous()
    dec = torch.full((N + 1, G), -1, dtype=torch.int32, device=dev)
    eng.decode_begin(first_token=int(codes[0, 0]), prefill_len=tie.shape[1], gen_step=0, past_hidden=hidden,
                     trailing_text=tth[0].to(dev).contiguous(), tts_pad_embed=tpe.view(-1).to(dev).contiguous(),
                     repetition_penalty=1.0, min_new_tokens=N, max_new_tokens=N, **greedy)
    eng.decode_set_forced(forced, dec)
    if graph:
        eng.graph_capture()
    else:
        eng.graph_reset()
    eng.decode_frames(N)
    n, _ = eng.decode_poll()
    assert n == N, (n, N)
    recorded = eng.decode_codes(0, N).cpu().numpy()
    assert np.array_equal(recorded, codes.astype(np.int64)), "the forced ids were not the ones the loop continued with"
    d = dec.cpu().numpy().astype(np.int64)
    out = d[:N].copy()
    out[0, 0] = int(tok0)                               # the prefill decision comes from the API sampler
    eng.decode_set_forced(None, None)
    return out


def near_ties(case: dict, k_ulp: float) -> int:
    """Number of the oracle's decisions whose top-2 margin is at most k_ulp bf16 ulps of the winning logit."""
    N = case["codes"].shape[0]
    margin = np.concatenate([case["t_margin"][:N, None], case["p_margin"]], axis=1)
    top1 = np.concatenate([case["t_top1"][:N, None], case["p_top1"]], axis=1)
    return int((margin / bf16_ulp(top1) <= k_ulp).sum())


def score(decisions: np.ndarray, case: dict, k_ulp: float):
    """-> dict(matched_decisions, total, matched_frames, frames, worst_mismatch_ulp, unexplained)"""
    codes = case["codes"].astype(np.int64)
    N = codes.shape[0]
    margin = np.concatenate([case["t_margin"][:N, None], case["p_margin"]], axis=1)        # [N, 16]
    top1 = np.concatenate([case["t_top1"][:N, None], case["p_top1"]], axis=1)
    m_ulp = margin / bf16_ulp(top1)
    bad = decisions != codes
    worst = float(m_ulp[bad].max()) if bad.any() else 0.0
    unexplained = int((bad & (m_ulp > k_ulp)).sum())
    return dict(total=int(codes.size), matched_decisions=int((~bad).sum()), frames=N,
                matched_frames=int((~bad).all(axis=1).sum()), worst_mismatch_ulp=round(worst, 3),
                smallest_margin_ulp=round(float(m_ulp.min()), 3), unexplained=unexplained, k_ulp=k_ulp)
