"""Parity at the BENCHMARKED configurations: full-depth 0.6B / 1.7B (28 talker + 5 predictor layers at the real
shapes), 200-token prompt (KV 200..224: four 64-key tiles, all attention workers), 24 greedy frames through the real
fused loop (hipGraph replay), fp32 AND bf16, every one of the 16 x 24 decisions scored by teacher forcing against
golden ids produced by the CPU oracle (oracle/make_golden_fulldepth.py -> tests/golden/fulldepth.npz).

Tolerances (north star: "bit-identical in RVQ token indices"):
  fp32 : every decision identical (the oracle's smallest top-2 margin in these vectors is > 1e-3, far above fp32
         summation-order noise).
  bf16 : every decision identical, except where the ORACLE's own top-2 margin is at most K_ULP = 3 bf16 ulps of the
         winning logit (bf16 logits are multiples of the ulp, so margins are 0, 1, 2, ... ulps) -- there the id is decided by
         summation order inside a dot product (HIP: fixed fma chains + DPP tree in the decode GEMVs, 8 K-slices x 2 MFMA chains
         in the short-prompt prefill GEMMs; CPU oracle: oneDNN blocking), not by the algorithm: every K / V cache entry is one
         bf16 rounding of such a sum, and 24 frames of logits sit on top of 200 x 28 of them.  That is the hard gate
         (`unexplained == 0`).  K_ULP history: the round-2 / early round-3 kernels happened to stay within 2 ulps on these
         vectors (worst mismatch exactly 2); with the weight-stationary prefill GEMMs and the vectorised row norm ONE decision
         at an oracle margin of 3 ulps falls the other way (0.6B), which is the bound the 8..32-lane MFMA batch path has been
         held to all along (tests/test_gpu_batch_fulldepth.py) -- the first-token logits of the two prefill variants differ by
         up to 0.05 = 3 ulps themselves (tools/prefill_time.py).
         The oracle ITSELF is no more reproducible than that: re-evaluated on one CPU thread, or with fp32 / fp64 operands, or
         with K summed in 8 slices, it reproduces 365-377 of its own 384 ids, worst flip at 2-3 ulps
         (oracle/selfcheck_fulldepth.py, tests/test_oracle_selfcheck.py).
         How many near-tie decisions fall the other way is a coin-flip statistic of the summation order, not a quality
         figure: the goldens hold 42 (0.6B) / 52 (1.7B) decisions with a margin <= 2 ulps out of 384 (9 / 13 exact ties).
         The second gate is a FROZEN floor on the number of identical decisions per shape (round 4; it replaces the
         round-3 "at most 30 % of the near-tie set flipped", which followed the measurement): the count the round-3 kernels
         measure on MI355X (deterministic: 0.6B 376, 1.7B 367 of 384) minus ONE decision -- MIN_MATCHED below -- so that a
         kernel that loses precision, and would flip more near-ties or any decision with a wider margin, fails.  In shares of
         the near-tie set that is <= 9 / 68 = 13 % (0.6B) and <= 18 / 73 = 25 % (1.7B; the <= 20 % the round-3 review asked for
         is not reachable there: the round-3 kernels already sit at 17 / 73, and the oracle re-evaluated with exactly rounded
         dot products flips 17 itself, oracle/selfcheck_fulldepth.py).  K_ULP and MIN_MATCHED do not move again: a kernel
         change that misses them is a regression of that kernel.  The figures are printed, written to
         gpurun_out/parity_fulldepth.json and carried into the bench line.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

K_ULP = 3.0                                     # frozen (round 3)
MIN_MATCHED = {"0p6b": 375, "1p7b": 366}        # frozen (round 4): identical decisions of 384, round-3 measurement minus one

from fq3hip.config import qwen3_tts_0p6b, qwen3_tts_1p7b
from fq3hip.weights import synth_weights, synth_prompt


def _note(key, val):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, "parity_fulldepth.json")
    cur = {}
    if os.path.exists(p):
        try:
            cur = json.load(open(p))
        except Exception:
            cur = {}
    cur[key] = val
    json.dump(cur, open(p, "w"), indent=1)


@pytest.mark.parametrize("size", ["0p6b", "1p7b"])
@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_full_depth_teacher_forced(size, tag, golden_dir):
    from fq3hip.engine import Fq3Engine
    from oracle import teacher_forced as TF
    g = np.load(os.path.join(golden_dir, "fulldepth.npz"))
    frames, plen, tlen = (int(x) for x in g["meta"])
    case = TF.load_case(g, f"{size}_{tag}")
    dtype = torch.float32 if tag == "f32" else torch.bfloat16
    cfg = qwen3_tts_0p6b() if size == "0p6b" else qwen3_tts_1p7b()
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    tie, tam, tth, tpe, _ = synth_prompt(cfg, plen, tlen, 0, dtype=dtype)
    eng = Fq3Engine(cfg, W, device="cuda", dtype=dtype, max_seq_len=plen + frames + 8, max_frames=frames + 8)
    del W
    eng.set_predictor_sampling(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    res = {}
    for graph in (True, False):
        dec = TF.forced_decisions(eng, cfg, tie, tth, tpe, case["codes"], graph=graph)
        s = TF.score(dec, case, K_ULP)
        res["graph" if graph else "direct"] = s
        print(f"[parity] {size} {tag} {'graph' if graph else 'direct'}: {s}")
        if tag == "f32":
            assert s["matched_decisions"] == s["total"], s
        else:
            assert s["unexplained"] == 0, s
            assert s["matched_decisions"] >= MIN_MATCHED[size], (s, TF.near_ties(case, K_ULP))
    assert res["graph"]["matched_decisions"] == res["direct"]["matched_decisions"]
    _note(f"{size}_{tag}", res["graph"])
    eng.close()


def test_free_running_bf16_prefix_matches(golden_dir):
    """Without forcing: the product loop's own greedy bf16 ids equal the oracle's up to the first near-tie decision
    (after which a greedy run legitimately follows another trajectory)."""
    from fq3hip.engine import Fq3Engine
    from oracle import teacher_forced as TF
    g = np.load(os.path.join(golden_dir, "fulldepth.npz"))
    frames, plen, tlen = (int(x) for x in g["meta"])
    case = TF.load_case(g, "0p6b_bf16")
    cfg = qwen3_tts_0p6b()
    dtype = torch.bfloat16
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    tie, tam, tth, tpe, _ = synth_prompt(cfg, plen, tlen, 0, dtype=dtype)
    eng = Fq3Engine(cfg, W, device="cuda", dtype=dtype, max_seq_len=plen + frames + 8, max_frames=frames + 8)
    eng.set_predictor_sampling(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    greedy = dict(temperature=1.0, top_k=0, top_p=1.0, do_sample=False)
    V = cfg.talker.vocab_size
    logits, hidden = eng.prefill(tie[0].cuda().contiguous())
    tok = eng.sample(logits, sup_lo=V - 1024, sup_hi=V, keep_id=cfg.codec_eos_token_id, suppress_eos=True, **greedy)
    eng.decode_begin(first_token=int(tok), prefill_len=plen, gen_step=0, past_hidden=hidden,
                     trailing_text=tth[0].cuda().contiguous(), tts_pad_embed=tpe.view(-1).cuda().contiguous(),
                     repetition_penalty=1.0, min_new_tokens=frames, max_new_tokens=frames, **greedy)
    eng.graph_capture()
    eng.decode_frames(frames)
    n, _ = eng.decode_poll()
    codes = eng.decode_codes(0, n).cpu().numpy()
    ref = case["codes"].astype(np.int64)
    same = (codes == ref).reshape(-1)
    first_bad = int(np.argmin(same)) if not same.all() else same.size
    margin = np.concatenate([case["t_margin"][:frames, None], case["p_margin"]], axis=1).reshape(-1)
    top1 = np.concatenate([case["t_top1"][:frames, None], case["p_top1"]], axis=1).reshape(-1)
    print(f"[parity] free-running bf16 0.6B: identical prefix {first_bad} of {same.size} decisions")
    _note("0p6b_bf16_free_running_prefix", [first_bad, int(same.size)])
    if first_bad < same.size:
        assert margin[first_bad] / TF.bf16_ulp(top1[first_bad:first_bad + 1])[0] <= K_ULP


def test_full_depth_sampled_fp32_exact(golden_dir):
    """Product-default SAMPLING (T 0.9, top-k 50, repetition penalty 1.05; predictor T 0.9 / top-k 50) at full depth and the
    real vocabulary sizes, fp32, with the oracle's pre-drawn Exp(1) noise: the free-running hipGraph loop must reproduce
    the oracle's ids exactly (golden: oracle/make_golden_fulldepth.py::run_sampled_case)."""
    from fq3hip.engine import Fq3Engine
    from oracle.make_golden_fulldepth import sampled_noise, SAMPLED_FRAMES, PROMPT, TRAILING
    g = np.load(os.path.join(golden_dir, "fulldepth.npz"))
    ref = g["0p6b_f32_sampled_codes"].astype(np.int64)
    cfg = qwen3_tts_0p6b()
    dtype = torch.float32
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    tie, tam, tth, tpe, _ = synth_prompt(cfg, PROMPT, TRAILING, 0, dtype=dtype)
    tn, pn = sampled_noise(cfg)
    frames = SAMPLED_FRAMES
    eng = Fq3Engine(cfg, W, device="cuda", dtype=dtype, max_seq_len=PROMPT + frames + 8, max_frames=frames + 8)
    del W
    eng.set_predictor_sampling(do_sample=True, top_k=50, top_p=1.0, temperature=0.9)
    kw = dict(temperature=0.9, top_k=50, top_p=1.0, do_sample=True)
    V = cfg.talker.vocab_size
    logits, hidden = eng.prefill(tie[0].cuda().contiguous())
    tok = eng.sample(logits, sup_lo=V - 1024, sup_hi=V, keep_id=cfg.codec_eos_token_id, suppress_eos=True, noise=tn[0].cuda().contiguous(), **kw)
    eng.decode_begin(first_token=int(tok), prefill_len=PROMPT, gen_step=0, past_hidden=hidden, trailing_text=tth[0].cuda().contiguous(),
                     tts_pad_embed=tpe.view(-1).cuda().contiguous(), repetition_penalty=1.05, min_new_tokens=frames,
                     max_new_tokens=frames, talker_noise=tn[1:].contiguous().cuda(), pred_noise=pn.contiguous().cuda(),
                     noise_frames=frames, **kw)
    eng.graph_capture()
    eng.decode_frames(frames)
    n, _ = eng.decode_poll()
    codes = eng.decode_codes(0, n).cpu().numpy()
    assert codes.shape == ref.shape and np.array_equal(codes, ref), int((codes != ref).sum())
