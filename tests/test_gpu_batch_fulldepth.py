"""Parity of the LOCK-STEP BATCH at the benchmarked configurations: full-depth 0.6B / 1.7B (28 talker + 5 predictor layers at
the real shapes), 8, 16, 32, 64 AND 128 lanes, the matrix-core batch GEMVs the bench line uses (the template instantiations selected by
H = 1024 / 2048, I = 3072 / 6144: gemv_batch_mfma_norm_kernel<8|16, *>, gemv_batch_mfma_plain_kernel<8|12|24, 8, *>), every lane
teacher-forced with golden oracle ids and every one of its 16 x frames decisions scored (oracle/teacher_forced.py).

Lanes alternate between two golden utterances (tests/golden/fulldepth.npz: 200-row prompt, 24 frames;
tests/golden/fulldepth_alt.npz: 137-row prompt, 16 frames -- other positions, a ragged last key tile), so the lanes of one
batch sit at different positions and finish at different frames.

Tolerances: bf16 -- a mismatch is accepted only where the ORACLE's own top-2 margin is <= K_ULP = 3 bf16 ulps of the winning
logit (the largest margin of any mismatch observed on this path: 2 ulps at the 0.6B shapes, 3 at the 1.7B shapes, whose
dot products are twice as long), and every lane must reach a FROZEN floor of identical decisions for its utterance: the
round-3 measurement per shape (profiles/r03_parity_batch_fulldepth.json) minus one decision per lane -- LANE_FLOOR below; it
replaces the round-3 "matched fraction >= 0.95", which 1.7B / 32 lanes cleared by a hair (0.9547).  K_ULP and the floors do
not move again.  Above 32 lanes (round 5) the floor of a NEW summation order comes from the oracle's own self-check, see below; the default
form's measured counts are frozen since round 6 (DEFAULT_FORM_FLOOR).  fp32 (VALU batch GEMVs, 32 lanes, 0.6B) -- every decision identical."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

K_ULP = 3.0                 # frozen (round 3)
# frozen (round 4): identical decisions per lane for (utterance A: 384 decisions, utterance B: 256), round-3 measurement minus one
LANE_FLOOR = {("0p6b", 8): (373, 240), ("0p6b", 16): (373, 240), ("0p6b", 32): (374, 245),
              ("1p7b", 8): (367, 248), ("1p7b", 16): (367, 248), ("1p7b", 32): (363, 246)}
# 64 and 128 lanes (round 4).  Above 32 lanes the normalising GEMVs run as ONE normalisation launch + the weight-stationary GEMM kernel
# (batch_kernels.cuh::rmsnorm_batch_kernel): the same normalised tokens bit for bit, another fp32 summation order of the products.  The
# floors of that form are its first measurement (378 / 243 and 368 / 248 per lane, at 64 and at 128 lanes alike) minus one decision
# per lane; the panel kernels' 5..8-tile instantiations are checked separately below against the 32-lane counts, which they must
# reproduce EXACTLY (a lane's arithmetic does not depend on the number of token tiles of the launch).
R4_FORM_FLOOR = {"0p6b": (377, 242), "1p7b": (367, 247)}
# Round 5: from 64 lanes the talker attention runs as one workgroup per (kv head, lane) that writes final head outputs
# (batch_kernels.cuh::attn_decode_lane_kernel: no partial slots, no merge launch), which changes the order in which keys enter the online
# softmax; and `norm_fused` 1 (a measured negative, off by default) moves the RMSNorm into the weight-stationary GEMM pair, which changes
# the fp32 order of the sum of squares.  A form with a summation order of its own does not get a floor cut from its own first measurement
# (the round-4 review): its floor is the ORACLE's reproducibility floor for the same utterance -- the smallest count any
# changed-accumulation re-evaluation of the oracle reaches against its own golden ids (tests/golden/fulldepth_selfcheck.json,
# oracle/selfcheck_fulldepth.py; CPU only) -- minus two.  The round-4 form (`attn_lane` 0, `norm_fused` 0) is re-run at 64 lanes and must
# still clear its frozen floors above.


# Round 6 (round-5 advisor): the DEFAULT path of 64 and 128 lanes (weight-stationary normalising GEMMs, lane attention, pair pass) has now
# been measured in two rounds with identical counts -- 373 / 245 (0.6B) and 366 / 248 (1.7B) per lane at 64 and at 128 lanes alike
# (profiles/r05_parity_batch_fulldepth.json, profiles/r06_parity_batch_fulldepth.json); every change of round 6 on that path (fragment-major
# weight copies, the group form of the predictor attention) is bit-identical by construction and asserted so decision by decision below.
# Those counts are FROZEN here, like the <= 32-lane ones: the default path must reach them exactly minus one decision per lane (and, as
# before, the oracle-derived floor, which a NEW summation order would be held to instead).
DEFAULT_FORM_FLOOR = {"0p6b": (372, 244), "1p7b": (365, 247)}


def oracle_floor(golden_dir, size):
    d = json.load(open(os.path.join(golden_dir, "fulldepth_selfcheck.json")))
    lo = lambda key: min(v["matched_decisions"] for k, v in d[key].items() if k != "native_bf16_again")
    return (lo(size) - 2, lo(size + "_alt") - 2)

from fq3hip.config import qwen3_tts_0p6b, qwen3_tts_1p7b
from fq3hip.weights import synth_weights, synth_prompt


def _note(key, val):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, "parity_batch_fulldepth.json")
    cur = {}
    if os.path.exists(p):
        try:
            cur = json.load(open(p))
        except Exception:
            cur = {}
    cur[key] = val
    json.dump(cur, open(p, "w"), indent=1)


def _cases(golden_dir, cfg, size, tag, dtype):
    """[(case dict, tie, tth, tpe)] for the two golden utterances of this model / dtype."""
    from oracle import teacher_forced as TF
    from oracle.make_golden_fulldepth_alt import alt_prompt
    g = np.load(os.path.join(golden_dir, "fulldepth.npz"))
    frames, plen, tlen = (int(x) for x in g["meta"])
    tie, _tam, tth, tpe, _ = synth_prompt(cfg, plen, tlen, 0, dtype=dtype)
    out = [(TF.load_case(g, f"{size}_{tag}"), tie, tth, tpe)]
    ga = np.load(os.path.join(golden_dir, "fulldepth_alt.npz"))
    if f"{size}_{tag}_codes" in ga:
        tie2, _tam2, tth2, tpe2, _ = alt_prompt(cfg, dtype)
        out.append((TF.load_case(ga, f"{size}_{tag}"), tie2, tth2, tpe2))
    return out


def _arm_forced(eng, cfg, case, tie, tth, tpe):
    """prefill + decode_begin with the oracle's first token + teacher forcing; returns (HIP prefill decision, forced, decisions)."""
    dev = eng.device
    codes = case["codes"]
    N, G = codes.shape
    V, eos = cfg.talker.vocab_size, cfg.codec_eos_token_id
    greedy = dict(temperature=1.0, top_k=0, top_p=1.0, do_sample=False)
    logits, hidden = eng.prefill(tie[0].to(dev).contiguous())
    tok0 = eng.sample(logits, sup_lo=max(0, V - 1024), sup_hi=V, keep_id=eos, suppress_eos=True, **greedy)
    forced = torch.zeros(N + 1, G, dtype=torch.int32)
    forced[:N] = torch.from_numpy(codes.astype(np.int32))
    forced[N, 0] = int(codes[N - 1, 0])
    forced = forced.to(dev).contiguous()
    dec = torch.full((N + 1, G), -1, dtype=torch.int32, device=dev)
    eng.decode_begin(first_token=int(codes[0, 0]), prefill_len=tie.shape[1], gen_step=0, past_hidden=hidden,
                     trailing_text=tth[0].to(dev).contiguous(), tts_pad_embed=tpe.view(-1).to(dev).contiguous(),
                     repetition_penalty=1.0, min_new_tokens=N, max_new_tokens=N, **greedy)
    eng.decode_set_forced(forced, dec)
    return int(tok0), forced, dec


def _run_batch(engines, cfg, cases, B, mfma, graph=True, options=()):
    from fq3hip.engine import Fq3Batch
    from oracle import teacher_forced as TF
    lanes = engines[:B]
    batch = Fq3Batch(lanes)
    batch.set_option("mfma", mfma)
    for k, v in options:
        batch.set_option(k, v)
    if os.environ.get("FQ3_TEST_NORM_SKINNY_ABOVE") is not None:      # development: where the weight-stationary form would start
        batch.set_option("norm_skinny_above", int(os.environ["FQ3_TEST_NORM_SKINNY_ABOVE"]))
    armed = []
    for i, e in enumerate(lanes):
        case, tie, tth, tpe = cases[i % len(cases)]
        armed.append((case,) + _arm_forced(e, cfg, case, tie, tth, tpe))
    if graph:
        batch.graph_capture()
    batch.frames(max(c[0]["codes"].shape[0] for c in cases))
    scores = []
    for i, (e, (case, tok0, forced, dec)) in enumerate(zip(lanes, armed)):
        N = case["codes"].shape[0]
        n, _ = e.decode_poll()
        assert n == N, (i, n, N)
        assert np.array_equal(e.decode_codes(0, N).cpu().numpy(), case["codes"].astype(np.int64)), f"lane {i}: the forced ids were not the ones the loop continued with"
        d = dec.cpu().numpy().astype(np.int64)[:N].copy()
        d[0, 0] = tok0
        scores.append(dict(TF.score(d, case, K_ULP), decisions=d))
        e.decode_set_forced(None, None)
    batch.close()
    return scores


@pytest.mark.parametrize("size", ["0p6b", "1p7b"])
def test_batch_full_depth_bf16_mfma_lanes_vs_oracle(size, golden_dir):
    from fq3hip.engine import Fq3Engine
    cfg = qwen3_tts_0p6b() if size == "0p6b" else qwen3_tts_1p7b()
    dtype = torch.bfloat16
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    cases = _cases(golden_dir, cfg, size, "bf16", dtype)
    seq = max(c[1].shape[1] + c[0]["codes"].shape[0] for c in cases) + 8
    frames = max(c[0]["codes"].shape[0] for c in cases) + 8
    first = Fq3Engine(cfg, W, device="cuda", dtype=dtype, max_seq_len=seq, max_frames=frames)
    del W
    engines = [first] + [Fq3Engine(cfg, first.weights, device="cuda", dtype=dtype, max_seq_len=seq, max_frames=frames, share=first) for _ in range(127)]
    for e in engines:
        e.set_predictor_sampling(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    per_lane_32 = None
    for B in (8, 16, 32, 64, 128):
        if B == 128:
            # the panel kernels' rolled-loop instantiations (five to eight tiles) at the real shapes: a lane's arithmetic does not depend on
            # the tile count, so its counts are EXACTLY those it has in the 32-lane batch
            panel = _run_batch(engines, cfg, cases, B, mfma=1, options=(("norm_skinny", 0), ("attn_lane", 0)))
            got = [s["matched_decisions"] for s in panel]
            _note(f"{size}_bf16_mfma_B{B}_panel_kernels", dict(per_lane=got))
            assert all(s["unexplained"] == 0 for s in panel)
            assert got == [per_lane_32[i % len(cases)] for i in range(B)], got
        if B == 64:
            # the round-4 form of this lane count (split-KV attention + merge launch; separate normalisation launch): bit-stable, frozen floors
            old = _run_batch(engines, cfg, cases, B, mfma=1, options=(("attn_lane", 0), ("norm_fused", 0)))
            _note(f"{size}_bf16_mfma_B{B}_round4_form", dict(per_lane=[s["matched_decisions"] for s in old]))
            assert all(s["unexplained"] == 0 for s in old)
            for i, sc in enumerate(old):
                assert sc["matched_decisions"] >= R4_FORM_FLOOR[size][i % len(cases)], (i, sc)
            # the RMSNorm folded into the GEMM pair (a measured negative, off by default, kept as a switch): correct all the same
            # the predictor's two-token prefill as ONE pass over 2 B rows (the default from 64 lanes) against two passes: every decision of
            # every lane identical (a row's arithmetic does not depend on the row count of a weight-stationary launch)
            two = _run_batch(engines, cfg, cases, B, mfma=1, options=(("pred_pair", 0),))
            one = _run_batch(engines, cfg, cases, B, mfma=1)
            for i, (a_, b_) in enumerate(zip(two, one)):
                assert np.array_equal(a_["decisions"], b_["decisions"]), f"lane {i}: the pair pass changed a decision"
            # the predictor attention per (kv group, lane) with live rows only (round 6 default) against per (q head, lane): bit-identical
            perhead = _run_batch(engines, cfg, cases, B, mfma=1, options=(("pred_attn_group", 0),))
            for i, (a_, b_) in enumerate(zip(perhead, one)):
                assert np.array_equal(a_["decisions"], b_["decisions"]), f"lane {i}: the group form of the predictor attention changed a decision"
            # the weight-stationary GEMMs on the fragment-major weight copies (round 6 default) against the row-major matrices: bit-identical
            rowmajor = _run_batch(engines, cfg, cases, B, mfma=1, options=(("packed_weights", 0),))
            for i, (a_, b_) in enumerate(zip(rowmajor, one)):
                assert np.array_equal(a_["decisions"], b_["decisions"]), f"lane {i}: the fragment-major weight copies changed a decision"
            fused = _run_batch(engines, cfg, cases, B, mfma=1, options=(("norm_fused", 1),))
            _note(f"{size}_bf16_mfma_B{B}_norm_fused", dict(per_lane=[s["matched_decisions"] for s in fused]))
            assert all(s["unexplained"] == 0 for s in fused)
            for i, sc in enumerate(fused):
                assert sc["matched_decisions"] >= oracle_floor(golden_dir, size)[i % len(cases)], (i, sc)
        scores = _run_batch(engines, cfg, cases, B, mfma=1)
        if B == 32:
            per_lane_32 = [s["matched_decisions"] for s in scores]
        tot = sum(s["total"] for s in scores); ok = sum(s["matched_decisions"] for s in scores)
        worst = max(s["worst_mismatch_ulp"] for s in scores)
        print(f"[parity] batch {size} bf16 mfma B={B}: {ok}/{tot} decisions identical, worst mismatch margin {worst} ulps, "
              f"per lane {[s['matched_decisions'] for s in scores]}")
        _note(f"{size}_bf16_mfma_B{B}", dict(matched=ok, total=tot, worst_mismatch_ulp=worst, k_ulp=K_ULP,
                                              per_lane=[s["matched_decisions"] for s in scores],
                                              unexplained=sum(s["unexplained"] for s in scores)))
        assert all(s["unexplained"] == 0 for s in scores), scores
        floor = LANE_FLOOR[(size, B)] if (size, B) in LANE_FLOOR else tuple(max(a_, b_) for a_, b_ in zip(DEFAULT_FORM_FLOOR[size], oracle_floor(golden_dir, size)))
        for i, sc in enumerate(scores):
            assert sc["matched_decisions"] >= floor[i % len(cases)], (i, sc, floor)
        # lanes that decode the same utterance must agree with each other exactly (lock-step lanes do not interact)
        for i in range(len(cases), B):
            assert scores[i]["matched_decisions"] == scores[i % len(cases)]["matched_decisions"]
    for e in engines[1:]:
        e.close()
    first.close()


def test_batch_full_depth_fp32_32_lanes_exact(golden_dir):
    """fp32 (VALU batch GEMVs, four LDS passes of 8 tokens each at 32 lanes): every decision of every lane identical to the oracle."""
    from fq3hip.engine import Fq3Engine
    cfg = qwen3_tts_0p6b()
    dtype = torch.float32
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    cases = _cases(golden_dir, cfg, "0p6b", "f32", dtype)
    seq = max(c[1].shape[1] + c[0]["codes"].shape[0] for c in cases) + 8
    frames = max(c[0]["codes"].shape[0] for c in cases) + 8
    first = Fq3Engine(cfg, W, device="cuda", dtype=dtype, max_seq_len=seq, max_frames=frames)
    del W
    engines = [first] + [Fq3Engine(cfg, first.weights, device="cuda", dtype=dtype, max_seq_len=seq, max_frames=frames, share=first) for _ in range(31)]
    for e in engines:
        e.set_predictor_sampling(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    scores = _run_batch(engines, cfg, cases, 32, mfma=0)
    print(f"[parity] batch 0p6b f32 B=32: per lane {[s['matched_decisions'] for s in scores]} of {[s['total'] for s in scores]}")
    _note("0p6b_f32_valu_B32", dict(matched=sum(s["matched_decisions"] for s in scores), total=sum(s["total"] for s in scores)))
    for s in scores:
        assert s["matched_decisions"] == s["total"], s
    for e in engines[1:]:
        e.close()
    first.close()
