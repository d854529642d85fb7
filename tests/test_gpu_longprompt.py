"""GPU: BASELINE configs[4] shape -- a 4096-token prompt at the real 1.7B layer dims (2 layers): matrix-core prefill with
the flash-style attention kernel, then decode steps over a > 4096-key cache, against golden vectors of the CPU oracle
(oracle/make_golden_longprompt.py).  fp32 contexts (exact-product kernels) to 2e-4 of the scale; bf16 to bf16 resolution
(0.025 x scale, the same bound as the short-prompt tests), with the flash kernel additionally held to the per-row wave
kernel (same inputs, fp32 softmax in both; only the summation order differs)."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fq3hip.weights import synth_weights, synth_prompt
from oracle.make_golden_longprompt import config, L


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_prefill_4096_and_decode_over_long_cache(tag, dtype, golden_dir):
    from fq3hip.engine import Fq3Engine
    g = np.load(os.path.join(golden_dir, "longprompt.npz"))
    cfg = config()
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    tie, tam, _, _, _ = synth_prompt(cfg, L, 4, 0, dtype=dtype)
    x = (tie * 30).to(dtype)[0].cuda().contiguous()
    eng = Fq3Engine(cfg, W, device="cuda", dtype=dtype, max_seq_len=L + 8, max_frames=8)
    tol = (lambda ref: 2e-4 * max(1.0, float(np.abs(ref).max()))) if dtype == torch.float32 else \
          (lambda ref: 0.025 * max(1.0, float(np.abs(ref).max())))
    results = {}
    for flash in ((1, 0) if dtype == torch.bfloat16 else (1,)):
        eng.set_option("flash_prefill", flash)
        eng.prefill(x)                                   # first call allocates the workspaces
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        logits, hidden = eng.prefill(x)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        lg, hd = logits.float().cpu().numpy(), hidden.float().cpu().numpy()
        assert np.abs(hd - g[f"hidden_{tag}"]).max() <= tol(g[f"hidden_{tag}"]), (tag, flash)
        assert np.abs(lg - g[f"logits_{tag}"]).max() <= tol(g[f"logits_{tag}"]), (tag, flash)
        results[flash] = (lg, hd, ms)
        print(f"[longprompt] {tag} flash={flash}: prefill({L} tokens, 2 layers at 1.7B dims) {ms:.1f} ms, "
              f"max |hidden - oracle| {np.abs(hd - g[f'hidden_{tag}']).max():.4f}")
        gen = torch.Generator().manual_seed(3)
        for step in range(2):
            xs = torch.randn(1, 1, cfg.talker.hidden_size, generator=gen).to(dtype)
            h = eng.talker_step(xs.view(-1).cuda(), L + step).float().cpu().numpy()
            assert np.abs(h - g[f"step{step}_{tag}"]).max() <= tol(g[f"step{step}_{tag}"]), (tag, flash, step)
    if dtype == torch.bfloat16:
        # flash (bf16 P split into high + residual) vs the wave kernel (fp32 P) after two layers: a few bf16 ulps apart at
        # most (measured: 2 ulps), and the flash result is at least as close to the oracle as the wave kernel's
        d = np.abs(results[1][1] - results[0][1]).max()
        assert d <= 2.0 ** -6 * max(1.0, float(np.abs(results[0][1]).max())), d
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "longprompt_prefill.txt"), "w") as f:
            f.write(f"prefill {L} tokens, 2 talker layers at 1.7B dims, bf16: flash MFMA attention {results[1][2]:.1f} ms, wave kernel {results[0][2]:.1f} ms\n")


def test_flash_prefill_left_padded_matches_wave_kernel():
    """n_pad > 0 across a tile boundary (n_pad = 70): padded keys are never attended, padded query rows are zeros."""
    from fq3hip.engine import Fq3Engine
    cfg = config()
    dtype = torch.bfloat16
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    tie, _, _, _, _ = synth_prompt(cfg, 300, 4, 0, dtype=dtype)
    x = (tie * 30).to(dtype)[0].cuda().contiguous()
    eng = Fq3Engine(cfg, W, device="cuda", dtype=dtype, max_seq_len=320, max_frames=8)
    outs = []
    for flash in (1, 0):
        eng.set_option("flash_prefill", flash)
        logits, hidden = eng.prefill(x, n_pad=70)
        outs.append((logits.float().cpu(), hidden.float().cpu()))
    assert (outs[0][1] - outs[1][1]).abs().max() <= 2.0 ** -7 * max(1.0, float(outs[1][1].abs().max()))
    assert (outs[0][0] - outs[1][0]).abs().max() <= 2.0 ** -6 * max(1.0, float(outs[1][0].abs().max()))


@pytest.mark.parametrize("Lp,n_pad", [(1500, 0), (1500, 70), (3200, 70), (3137, 0)])
def test_flash_prefill_paired_block_variants_match_wave_kernel(Lp, n_pad):
    """The mid-length (>= 1024 tokens: 64-query blocks handled in pairs bx / nqb-1-bx) and long (>= 3072: 128-query blocks, 8
    waves, pairs; 3137 tokens = an odd block count with a ragged last block) shapes of the flash kernel against the per-row wave
    kernel on the same inputs.  Compared on EVERY row: the second layer's K / V cache rows are functions of the first layer's
    attention output of that row, the last row's hidden / logits of all of them."""
    from fq3hip.engine import Fq3Engine
    cfg = config()
    dtype = torch.bfloat16
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    tie, _, _, _, _ = synth_prompt(cfg, Lp, 4, 0, dtype=dtype)
    x = (tie * 30).to(dtype)[0].cuda().contiguous()
    eng = Fq3Engine(cfg, W, device="cuda", dtype=dtype, max_seq_len=Lp + 8, max_frames=8)
    outs = []
    for flash in (1, 0):
        eng.set_option("flash_prefill", flash)
        logits, hidden = eng.prefill(x, n_pad=n_pad)
        k, v = eng.kv_export(cfg.talker.num_hidden_layers - 1, Lp)
        outs.append((logits.float().cpu(), hidden.float().cpu(), k.float().cpu()[:, n_pad:], v.float().cpu()[:, n_pad:]))
    for i, name in enumerate(("logits", "hidden", "K of the last layer", "V of the last layer")):
        a, b = outs[0][i], outs[1][i]
        d = float((a - b).abs().max())
        assert d <= 2.0 ** -6 * max(1.0, float(b.abs().max())), (name, d)
    # all rows really differ from zero (a skipped query block would leave stale / zero rows behind)
    assert float(outs[0][3].abs().amax(dim=(0, 2)).min()) > 0


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_config4_full_depth_prefill_and_teacher_forced_frames(tag, dtype, golden_dir):
    """BASELINE configs[4] at the depth it is benchmarked at (/root/reference/faster_qwen3_tts/model.py:1328-1505 +
    generate.py:107-134): all 28 talker layers + 5 predictor layers at the 1.7B shapes, a 4096-token prompt, then 8 greedy
    frames over the > 4096-key cache through the real fused loop (hipGraph replay), every one of the 16 x 8 decisions scored
    by teacher forcing against the CPU oracle's golden ids (oracle/make_golden_longprompt_full.py ->
    tests/golden/longprompt_full.npz).  fp32: prefill outputs to 2e-4 of the scale (measured 5e-6), every decision identical.
    bf16: prefill outputs to bf16 resolution after 28 layers (0.04 x scale; measured 0.023), decisions under the near-tie rule of
    tests/test_gpu_fulldepth.py with the floor THIS shape has: no mismatch where the oracle's own top-2 margin exceeds
    K_ULP_4096 = 4 bf16 ulps of the winning logit, and at least MIN_MATCHED_4096 identical decisions.  Why 4 and not the 3 of the
    200-token goldens: the oracle re-evaluated on these very goldens with nothing changed but the accumulation inside its dot
    products (oracle/selfcheck_fulldepth.py 1p7b_4096 -> tests/golden/fulldepth_selfcheck.json, pinned by
    tests/test_oracle_selfcheck.py) flips 2-6 of its own 128 decisions, and its `fp32_operands_ksplit8` variant -- K summed in 8
    slices, the shape of the HIP GEMMs' accumulation -- flips one at an oracle margin of exactly 4 ulps; 4096 keys of attention and
    28 layers sit under every logit.  Both figures were fixed when the test was first measured (round 4: 122 / 128, worst 4 ulps)
    and do not move."""
    import json
    from fq3hip.config import qwen3_tts_1p7b
    from fq3hip.engine import Fq3Engine
    from oracle import teacher_forced as TF
    g = np.load(os.path.join(golden_dir, "longprompt_full.npz"))
    frames, plen, tlen = (int(x) for x in g["meta"])
    case = TF.load_case(g, f"1p7b_{tag}")
    cfg = qwen3_tts_1p7b()
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    tie, tam, tth, tpe, _ = synth_prompt(cfg, plen, tlen, 0, dtype=dtype)
    eng = Fq3Engine(cfg, W, device="cuda", dtype=dtype, max_seq_len=plen + frames + 8, max_frames=frames + 8)
    del W
    eng.set_predictor_sampling(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    logits, hidden = eng.prefill(tie[0].cuda().contiguous())
    lg, hd = logits.float().cpu().numpy(), hidden.float().cpu().numpy()
    ref_l, ref_h = g[f"1p7b_{tag}_logits"], g[f"1p7b_{tag}_hidden"]
    rel = 2e-4 if dtype == torch.float32 else 0.04
    d_h = float(np.abs(hd - ref_h).max()) / max(1.0, float(np.abs(ref_h).max()))
    d_l = float(np.abs(lg - ref_l).max()) / max(1.0, float(np.abs(ref_l).max()))
    print(f"[config4 full depth] {tag}: max |hidden - oracle| / scale {d_h:.2e}, logits {d_l:.2e}")
    assert d_h <= rel and d_l <= rel, (tag, d_h, d_l)
    K_ULP = 4.0                 # K_ULP_4096, frozen (see the docstring)
    MIN_MATCHED_4096 = 121      # of 128, frozen: the first measurement (122) minus one
    res = {}
    for graph in (True, False):
        dec = TF.forced_decisions(eng, cfg, tie, tth, tpe, case["codes"], graph=graph)
        s = TF.score(dec, case, K_ULP)
        res[graph] = s
        print(f"[config4 full depth] {tag} {'graph' if graph else 'direct'}: {s}")
        if tag == "f32":
            assert s["matched_decisions"] == s["total"], s
        else:
            assert s["unexplained"] == 0 and s["matched_decisions"] >= MIN_MATCHED_4096, s
    assert res[True]["matched_decisions"] == res[False]["matched_decisions"]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    p = os.path.join(out, "parity_config4_fulldepth.json")
    cur = json.load(open(p)) if os.path.exists(p) else {}
    cur[tag] = dict(res[True], hidden_rel_err=d_h, logits_rel_err=d_l, near_ties=TF.near_ties(case, K_ULP))
    json.dump(cur, open(p, "w"), indent=1)
    eng.close()
