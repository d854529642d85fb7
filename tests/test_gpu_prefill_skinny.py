"""GPU: the weight-stationary GEMMs of the short-prompt prefill (csrc/skinny_gemm.cuh: prompts of <= 416 rows at the real
talker widths, K in {1024, 2048, 3072, 6144}) -- qkv, the fused [gate | up] + SwiGLU launch, o_proj / down with the residual.

Two talker layers at the 0.6B and at the 1.7B dims.  Checked against (a) the CPU oracle run here on the same bf16-valued
weights with fp32 arithmetic (0.025 x scale: the bound of the other bf16 prefill tests) and (b) the tiled / split-K kernels the
same library runs with `fq3_set_option("skinny_gemm", 0)` -- same inputs, only the fp32 summation order differs, so a few
bf16 ulps after two layers (2^-6 x scale) -- on the last row's hidden state and logits AND on every row of the last layer's
K / V cache (a skipped token tile or weight-row block would leave a stale row or column behind).  Prompt lengths: 200 (the
benchmarked one: 13 token tiles, the last one ragged), 37 (three tiles: fewer tiles than ring stages), 416 (the largest
served), 16 (one tile); a left-padded prompt; and a packed prefill of two prompts (350 packed rows) against the two single
prefills."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fq3hip.config import qwen3_tts_0p6b, qwen3_tts_1p7b
from fq3hip.weights import synth_weights, synth_prompt
from oracle import qwen3tts_oracle as O


def _cfg(size):
    cfg = qwen3_tts_0p6b() if size == "0p6b" else qwen3_tts_1p7b()
    cfg.talker.num_hidden_layers = 2
    cfg.predictor.num_hidden_layers = 1
    return cfg


def _setup(size, L, max_seq=None):
    from fq3hip.engine import Fq3Engine
    cfg = _cfg(size)
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("talker", "predictor"))
    tie, tam, _, _, _ = synth_prompt(cfg, L, 4, 0, dtype=torch.bfloat16)
    tie = (tie * 30).to(torch.bfloat16)                          # O(1) activations
    eng = Fq3Engine(cfg, W, device="cuda", dtype=torch.bfloat16, max_seq_len=max_seq or L + 8, max_frames=8)
    return cfg, W, tie, tam, eng


def _run(eng, cfg, x, L, n_pad=0):
    logits, hidden = eng.prefill(x, n_pad=n_pad)
    k, v = eng.kv_export(cfg.talker.num_hidden_layers - 1, L)
    return logits.float().cpu(), hidden.float().cpu(), k.float().cpu()[:, n_pad:], v.float().cpu()[:, n_pad:]


@pytest.mark.parametrize("size", ["0p6b", "1p7b"])
@pytest.mark.parametrize("L", [200, 37, 416, 16])
def test_skinny_prefill_matches_oracle_and_tiled_kernels(size, L):
    cfg, W, tie, tam, eng = _setup(size, L)
    x = tie[0].cuda().contiguous()
    eng.set_option("skinny_gemm", 1)
    got = _run(eng, cfg, x, L)
    eng.set_option("skinny_gemm", 0)
    ref = _run(eng, cfg, x, L)
    for i, name in enumerate(("logits", "hidden", "K of the last layer", "V of the last layer")):
        d = float((got[i] - ref[i]).abs().max())
        assert d <= 2.0 ** -6 * max(1.0, float(ref[i].abs().max())), (name, d)
    assert float(got[3].abs().amax(dim=(0, 2)).min()) > 0                  # every cache row written
    # the oracle: fp32 arithmetic on the same bf16-valued weights
    Wq = {k: v.float() for k, v in W.items()}
    orc = O.OracleTTS(cfg, Wq, max_seq_len=L + 8)
    with torch.inference_mode():
        lo, ho, _, _ = orc.prefill(tie.float(), tam)
    lo, ho = lo.float(), ho.float().view(-1)
    assert float((got[1] - ho).abs().max()) <= 0.025 * max(1.0, float(ho.abs().max()))
    assert float((got[0] - lo).abs().max()) <= 0.025 * max(1.0, float(lo.abs().max()))
    print(f"[skinny prefill] {size} L={L}: max |hidden - oracle| {float((got[1] - ho).abs().max()):.4f} (tiled kernels: "
          f"{float((ref[1] - ho).abs().max()):.4f}), skinny vs tiled {float((got[1] - ref[1]).abs().max()):.4f}")


@pytest.mark.parametrize("size", ["0p6b", "1p7b"])
@pytest.mark.parametrize("L", [200, 37])
def test_fragment_major_weight_copies_are_bit_identical(size, L):
    """Round 6: fq3_bind_weights keeps a FRAGMENT-MAJOR copy of every bf16 layer matrix (the kilobyte an A-operand load of one wave
    reads is contiguous; csrc/skinny_gemm.cuh, SkinnyArgs::Wp) and the weight-stationary GEMMs read it ("packed_weights", default 1).
    The same values reach the same registers, so the prefill's logits, hidden state and K / V rows are those of the row-major
    matrices bit for bit -- plain 16-row blocks (qkv, o_proj, down) and the [gate | up] pairing alike."""
    cfg, W, tie, tam, eng = _setup(size, L)
    x = tie[0].cuda().contiguous()
    eng.set_option("packed_weights", 1)
    got = _run(eng, cfg, x, L)
    eng.set_option("packed_weights", 0)
    ref = _run(eng, cfg, x, L)
    for i, name in enumerate(("logits", "hidden", "K of the last layer", "V of the last layer")):
        assert torch.equal(got[i], ref[i]), name
    assert float(got[3].abs().amax(dim=(0, 2)).min()) > 0


def test_skinny_prefill_left_padded():
    L, n_pad = 200, 5
    cfg, W, tie, tam, eng = _setup("0p6b", L)
    x = tie[0].cuda().contiguous()
    outs = []
    for v in (1, 0):
        eng.set_option("skinny_gemm", v)
        outs.append(_run(eng, cfg, x, L, n_pad=n_pad))
    for i in range(4):
        d = float((outs[0][i] - outs[1][i]).abs().max())
        assert d <= 2.0 ** -6 * max(1.0, float(outs[1][i].abs().max())), (i, d)


def test_packed_prefill_of_two_short_prompts_through_the_skinny_kernels():
    """150 + 200 packed rows (<= 416: every row-wise GEMM of the packed pass is a weight-stationary launch) against the two
    single prefills: the rows of a prompt see the same arithmetic either way except for which token tile they sit in."""
    from fq3hip.engine import Fq3Engine
    cfg = _cfg("0p6b")
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("talker", "predictor"))
    engs = [Fq3Engine(cfg, W, device="cuda", dtype=torch.bfloat16, max_seq_len=512, max_frames=8)]
    engs.append(Fq3Engine(cfg, W, device="cuda", dtype=torch.bfloat16, max_seq_len=512, max_frames=8, share=engs[0]))
    xs = []
    for L, seed in ((150, 1), (200, 2)):
        tie, _, _, _, _ = synth_prompt(cfg, L, 4, seed, dtype=torch.bfloat16)
        xs.append((tie * 30).to(torch.bfloat16)[0].cuda().contiguous())
    single = []
    for e, x in zip(engs, xs):
        lg, hd = e.prefill(x)
        k, v = e.kv_export(1, x.shape[0])
        single.append((lg.float().cpu(), hd.float().cpu(), k.float().cpu(), v.float().cpu()))
    packed = Fq3Engine.prefill_batch(engs, xs)
    for q, (e, x) in enumerate(zip(engs, xs)):
        lg, hd = packed[q]
        k, v = e.kv_export(1, x.shape[0])
        got = (lg.float().cpu(), hd.float().cpu(), k.float().cpu(), v.float().cpu())
        for i in range(4):
            d = float((got[i] - single[q][i]).abs().max())
            assert d <= 2.0 ** -6 * max(1.0, float(single[q][i].abs().max())), (q, i, d)


@pytest.mark.parametrize("L,n_pad", [(200, 0), (37, 0), (256, 0), (16, 0), (200, 5), (130, 70)])
def test_resident_tile_flash_prefill_is_bit_identical(L, n_pad):
    """Prompts of <= 256 rows (round 5): flash_prefill_small_kernel keeps every key tile of a query block in LDS (one barrier) instead of
    streaming them through one stage (two barriers + a global round trip per tile).  Per wave the arithmetic is the same function
    (flash_tile) over the same tiles in the same order: logits, hidden state and every K / V row are IDENTICAL to the streamed-tile
    kernel -- whole tiles, a ragged last tile, a single tile, left padding inside the first tile and across a tile boundary."""
    cfg, W, tie, tam, eng = _setup("0p6b", L)
    x = tie[0].cuda().contiguous()
    outs = []
    for v in (1, 0):
        eng.set_option("flash_small", v)
        outs.append(_run(eng, cfg, x, L, n_pad=n_pad))
    for i, name in enumerate(("logits", "hidden", "K", "V")):
        assert torch.equal(outs[0][i], outs[1][i]), name


def test_packed_prefill_attention_in_one_launch_per_layer():
    """fq3_prefill_batch over contexts of ONE pool, every prompt <= 256 rows: q / k norm + RoPE + KV write and the causal attention of
    ALL prompts run as two launches per layer (blockIdx.z = sequence) instead of two per prompt and layer.  Against the per-prompt
    launches (`flash_small` 0) of the same packed pass: bit-identical (the GEMMs are the same launches either way); against the single
    prefills: the packed-GEMM tolerance of the test above."""
    from fq3hip.engine import Fq3Engine, Fq3KvPool
    cfg = _cfg("0p6b")
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("talker", "predictor"))
    first = Fq3Engine(cfg, W, device="cuda", dtype=torch.bfloat16, max_seq_len=512, max_frames=8)
    pool = Fq3KvPool(cfg, 24, dtype=torch.bfloat16)
    engs = [Fq3Engine(cfg, W, device="cuda", dtype=torch.bfloat16, max_seq_len=512, max_frames=8, share=first, pool=pool) for _ in range(3)]
    specs = ((150, 1, 0), (200, 2, 3), (64, 3, 0))
    xs, pads = [], []
    for L, seed, n_pad in specs:
        tie, _, _, _, _ = synth_prompt(cfg, L, 4, seed, dtype=torch.bfloat16)
        xs.append((tie * 30).to(torch.bfloat16)[0].cuda().contiguous())
        pads.append(n_pad)

    def snapshot(results):
        out = []
        for e, x, p, (lg, hd) in zip(engs, xs, pads, results):
            k, v = e.kv_export(1, x.shape[0])
            out.append((lg.float().cpu(), hd.float().cpu(), k.float().cpu()[:, p:], v.float().cpu()[:, p:]))
        return out

    single = []
    for e, x, p in zip(engs, xs, pads):
        e.set_option("flash_small", 0)
        single.append(e.prefill(x, n_pad=p))
    single = snapshot(single)
    packed = {}
    for mode in (1, 0):
        for e in engs:
            e.set_option("flash_small", mode)
        packed[mode] = snapshot(Fq3Engine.prefill_batch(engs, xs, n_pads=pads))
    for q in range(3):
        for i in range(4):
            assert torch.equal(packed[1][q][i], packed[0][q][i]), (q, i)
            d = float((packed[1][q][i] - single[q][i]).abs().max())
            assert d <= 2.0 ** -6 * max(1.0, float(single[q][i].abs().max())), (q, i, d)
    for e in engs + [first]:
        e.close()


@pytest.mark.parametrize("size", ["0p6b", "1p7b"])
@pytest.mark.parametrize("L", [1000, 517])
def test_swiglu_in_the_ring_tile_is_bit_identical(size, L):
    """Round 6: a many-row prefill (> 416 rows: packed prefills, long prompts) runs gate | up on the 256-wide ring tile over a copy of the
    weight whose halves are interleaved in 16-row blocks, with SwiGLU in the tile's epilogue ("swiglu_tile", default 1; GemmArgs::Wi) --
    the [M][2I] image and the elementwise launch are gone.  y = rnd(rnd(silu(rnd(g))) * rnd(u)) either way: logits, hidden state and the
    K / V rows of the last layer equal those of the GEMM + silu_mul pair bit for bit (1000 rows: whole tiles; 517: a ragged last tile)."""
    cfg, W, tie, tam, eng = _setup(size, L)
    x = tie[0].cuda().contiguous()
    eng.set_option("swiglu_tile", 1)
    got = _run(eng, cfg, x, L)
    eng.set_option("swiglu_tile", 0)
    ref = _run(eng, cfg, x, L)
    for i, name in enumerate(("logits", "hidden", "K of the last layer", "V of the last layer")):
        assert torch.equal(got[i], ref[i]), name
    assert float(got[3].abs().amax(dim=(0, 2)).min()) > 0
