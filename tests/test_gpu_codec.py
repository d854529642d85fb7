"""GPU parity of the HIP 12 Hz codec decoder (through the C ABI) vs the CPU oracle / golden vectors.

Tolerance (north star: PCM RMS error <= 1e-3 vs the Torch path): fp32 context <= 1e-4 RMS (summation
order only); bf16 context is compared with the bf16 oracle (same rounding points) and must stay
within 1e-2 RMS / relative 5 % of the signal RMS -- bf16 rounding flips of single activations are
amplified by the un-normalised synthetic conv trunk; the fp32 bound is the kernel-correctness gate.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fq3hip.config import tiny_test_config
from fq3hip.weights import synth_weights


def _rms(x):
    return float(np.sqrt(np.mean(np.square(x.astype(np.float64)))))


@pytest.mark.parametrize("dtype,tag,tol", [(torch.float32, "f32", 1e-4), (torch.bfloat16, "bf16", 2e-2)])
def test_codec_decode_golden(dtype, tag, tol, golden_dir):
    from fq3hip.codec import HipSpeechTokenizer
    g = np.load(os.path.join(golden_dir, "codec.npz"))
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, dtype, parts=("codec",))
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", dtype, max_frames=64)
    codes = torch.from_numpy(g["codes"])
    wavs, sr = tok.decode({"audio_codes": codes.unsqueeze(0).cuda()})
    assert sr == 24000 and len(wavs) == 1
    wav = wavs[0].float().cpu().numpy()
    ref = g[f"wav_{tag}"]
    assert wav.shape == ref.shape == (tok.num_samples(codes.shape[0]),)
    err = _rms(wav - ref)
    assert err <= tol, (tag, err, _rms(ref))
    assert np.abs(wav).max() <= 1.0


@pytest.mark.parametrize("T", [1, 2, 7, 40])
def test_codec_lengths_and_oracle_fp32(T):
    from fq3hip.codec import HipSpeechTokenizer
    from oracle import qwen3tts_oracle as O
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("codec",))
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", torch.float32, max_frames=64)
    g = torch.Generator().manual_seed(100 + T)
    codes = torch.randint(0, cfg.codec.codebook_size, (T, cfg.codec.num_quantizers), generator=g)
    ref = O.codec_decode(codes, W, cfg.codec).numpy()
    wav = tok.decode_tensor(codes.cuda()).cpu().numpy()
    assert wav.shape == ref.shape
    assert _rms(wav - ref) <= 1e-4
    # causality: a prefix of the codes gives a prefix of the waveform (what the streaming windowing relies on)
    if T >= 7:
        w2 = tok.decode_tensor(codes[:5].cuda()).cpu().numpy()
        assert _rms(w2 - wav[: len(w2)]) <= 1e-4


def test_codec_single_piece_limit_is_an_abi_error():
    """The C entry point refuses more frames than its workspace holds (the host wrapper never asks for that)."""
    from fq3hip.codec import HipSpeechTokenizer
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("codec",))
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", torch.bfloat16, max_frames=8)
    with pytest.raises(RuntimeError):
        tok._decode_piece(torch.zeros(9, 16, dtype=torch.long, device="cuda"))


def test_codec_longer_than_workspace_is_chunked_like_upstream():
    """T > max_frames: decoded piecewise with a 25-frame left context, the upstream ``chunked_decode``
    scheme (transformers sibling modeling_qwen3_omni_moe.py:3686-3696), equal to the oracle's restatement of it."""
    from fq3hip.codec import HipSpeechTokenizer
    from oracle import qwen3tts_oracle as O
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("codec",))
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", torch.float32, max_frames=64)
    tok.CHUNK_FRAMES = 39                                                               # pieces of 39 new + 25 context frames
    g = torch.Generator().manual_seed(9)
    codes = torch.randint(0, cfg.codec.codebook_size, (100, cfg.codec.num_quantizers), generator=g)
    ref = O.codec_chunked_decode(codes, W, cfg.codec, chunk_size=39, left_context_size=25).numpy()
    wavs, _ = tok.decode({"audio_codes": codes.unsqueeze(0).cuda()})
    wav = wavs[0].cpu().numpy()
    assert wav.shape == ref.shape
    assert _rms(wav - ref) <= 1e-4


# ---- the shapes bench.py runs (latent 1024, decoder_dim 1536, 8 layers, head_dim 64, window 72) -------------------
def _real_codec(dtype, max_frames=208):
    from fq3hip.codec import HipSpeechTokenizer
    from fq3hip.config import qwen3_tts_0p6b
    cfg = qwen3_tts_0p6b()
    W = synth_weights(cfg, 0, dtype, parts=("codec",), codec_normalized=True)
    return cfg, HipSpeechTokenizer(cfg.codec, W, "cuda", dtype, max_frames=max_frames)


@pytest.mark.parametrize("T", [40, 100])
def test_codec_real_shapes_fp32(T, golden_dir):
    """fp32 context vs the fp32 oracle golden at the real shapes: RMS <= 1e-4 (north-star bound 1e-3).  T = 100 > the
    sliding window (72), head_dim 64, decoder_dim 1536."""
    g = np.load(os.path.join(golden_dir, "codec_real.npz"))
    cfg, tok = _real_codec(torch.float32)
    codes = torch.from_numpy(g[f"codes_{T}"].astype(np.int64))
    wav = tok.decode_tensor(codes.cuda()).cpu().numpy()
    ref = g[f"pcm_f32_{T}"]
    assert wav.shape == ref.shape
    err = _rms(wav - ref)
    print(f"[parity] codec real shapes fp32 T={T}: RMS {err:.3e} (signal RMS {_rms(ref):.3f})")
    assert err <= 1e-4, err


@pytest.mark.parametrize("T", [40, 100])
def test_codec_real_shapes_bf16(T, golden_dir):
    """bf16 context at the real shapes, measured against BOTH oracles.  The north star's 1e-3 is not reachable by any
    bf16 evaluation of this network: the oracle's own bf16 run is 8.4e-3 RMS (5 % of the signal) from its fp32 run on
    these normalised weights (oracle/make_golden_codec_real.py).  The gate is therefore relative to that floor: the HIP
    bf16 waveform must be no further from the fp32 truth than 1.25 x the oracle's bf16 waveform is, and closer to the
    bf16 oracle (same rounding points) than the two oracles are to each other."""
    g = np.load(os.path.join(golden_dir, "codec_real.npz"))
    cfg, tok = _real_codec(torch.bfloat16)
    codes = torch.from_numpy(g[f"codes_{T}"].astype(np.int64))
    wav = tok.decode_tensor(codes.cuda()).cpu().numpy()
    ref32 = g[f"pcm_f32_{T}"]
    refb = torch.from_numpy(g[f"pcm_bf16bits_{T}"]).view(torch.bfloat16).float().numpy()
    floor = _rms(refb - ref32)
    d32, db = _rms(wav - ref32), _rms(wav - refb)
    print(f"[parity] codec real shapes bf16 T={T}: HIP-vs-fp32-oracle {d32:.3e}, HIP-vs-bf16-oracle {db:.3e}, "
          f"oracle bf16-vs-fp32 floor {floor:.3e}")
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"parity_codec_bf16_T{T}.txt"), "w") as f:
        f.write(f"hip_vs_f32 {d32:.4e}\nhip_vs_bf16 {db:.4e}\noracle_floor {floor:.4e}\n")
    assert d32 <= 1.25 * floor and db <= floor, (d32, db, floor)


@pytest.mark.parametrize("T", [40, 100])
def test_codec_high_precision_mode_meets_1e_3(T, golden_dir):
    """codec_precision="fp32" on a bf16 weight table (what a bf16 checkpoint is): the weights are widened exactly, activations
    and matrix-core products are fp32.  Against the fp32-ARITHMETIC oracle on the same bf16-valued weights
    (tests/golden/codec_real_q.npz) the waveform must be within the north star's 1e-3 RMS -- measured ~1e-6 -- while the bf16
    arithmetic of the same weights is 7.6e-3 away from it (the oracle's own bf16 run: oracle/make_golden_codec_real.py)."""
    from fq3hip.codec import HipSpeechTokenizer
    from fq3hip.config import qwen3_tts_0p6b
    g = np.load(os.path.join(golden_dir, "codec_real_q.npz"))
    cfg = qwen3_tts_0p6b()
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("codec",), codec_normalized=True)          # the "checkpoint": bf16
    codes = torch.from_numpy(g[f"codes_{T}"].astype(np.int64)).cuda()
    ref = g[f"pcm_f32q_{T}"]
    hp = HipSpeechTokenizer(cfg.codec, W, "cuda", torch.float32, max_frames=208)
    lo = HipSpeechTokenizer(cfg.codec, W, "cuda", torch.bfloat16, max_frames=208)
    wav_hp, wav_lo = hp.decode_tensor(codes).cpu().numpy(), lo.decode_tensor(codes).cpu().numpy()
    e_hp, e_lo = _rms(wav_hp - ref), _rms(wav_lo - ref)
    print(f"[parity] codec T={T} vs fp32-arithmetic oracle on the bf16 checkpoint weights: fp32 mode {e_hp:.3e}, bf16 mode {e_lo:.3e} "
          f"(signal RMS {_rms(ref):.3f})")
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"parity_codec_hp_T{T}.txt"), "w") as f:
        f.write(f"fp32_mode_vs_f32q {e_hp:.4e}\nbf16_mode_vs_f32q {e_lo:.4e}\n")
    assert wav_hp.shape == ref.shape and e_hp <= 1e-4, e_hp            # two orders inside the north-star bound
    assert e_lo <= 1.25 * 7.6e-3                                         # the bf16 mode stays at the bf16 arithmetic floor
    hp.close(); lo.close()


@pytest.mark.parametrize("precision", ["bf16", "bf16x2"])
def test_codec_fused_residual_units_are_bit_identical(precision):
    """resunit_kernel (blocks with 192 / 96 channels: conv1 k7 -> SnakeBeta -> 1x1 conv -> + skip -> SnakeBeta in one launch, the middle
    tensor in LDS) against the two-GEMM path: the waveform is identical bit for bit, for a full decode (T = 40 and T = 100 > window)
    and for tail decodes (the row ranges of the fused op follow the conv1 halo).  bf16 x 2 (round 5): the 96-channel block's units,
    `mid` parked in LDS as (hi | lo) words (mode 2 then equals mode 1: its 192-channel units stay on two GEMMs)."""
    from fq3hip.codec import HipSpeechTokenizer
    from fq3hip.config import qwen3_tts_0p6b
    if precision == "bf16":
        cfg, tok = _real_codec(torch.bfloat16)
    else:
        cfg = qwen3_tts_0p6b()
        W = synth_weights(cfg, 0, torch.bfloat16, parts=("codec",), codec_normalized=True)
        tok = HipSpeechTokenizer(cfg.codec, W, "cuda", torch.bfloat16, max_frames=208, precision="bf16x2")
    g = torch.Generator().manual_seed(11)
    for T in (40, 100):
        codes = torch.randint(0, cfg.codec.codebook_size, (T, cfg.codec.num_quantizers), generator=g).cuda()
        tok.set_option("fuse_units", 0)
        plain = tok.decode_tensor(codes)
        n = plain.numel()
        for mode in (1, 2):                                  # 1: the 96-channel block, 2: the 192-channel block too
            tok.set_option("fuse_units", mode)
            assert torch.equal(tok.decode_tensor(codes), plain), (T, mode)
            for first in (n - 8 * 1920, n // 2 + 7):
                assert torch.equal(tok.decode_tensor(codes, first), plain[first:]), (T, mode, first)
    tok.set_option("fuse_units", 0)
    tok.close()


def test_codec_real_shapes_causal_prefix_property():
    """Size-independent property at the benchmark's full length (T = 200, fp32 and bf16): the decoder is causal, so the
    waveform of a prefix of the codes is BIT-IDENTICAL to the prefix of the waveform (what streaming relies on)."""
    for dtype in (torch.float32, torch.bfloat16):
        cfg, tok = _real_codec(dtype)
        g = torch.Generator().manual_seed(77)
        codes = torch.randint(0, cfg.codec.codebook_size, (200, cfg.codec.num_quantizers), generator=g).cuda()
        full = tok.decode_tensor(codes)
        for n in (73, 120):
            part = tok.decode_tensor(codes[:n].contiguous())
            assert torch.equal(part, full[: part.numel()]), (dtype, n)
        assert float(full.abs().max()) <= 1.0 and float(full.std()) > 0.05
        tok.close()



@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_codec_tail_decode_is_bit_identical_tiny(dtype):
    """fq3_codec_decode_tail == the tail of fq3_codec_decode, bit for bit, for every kind of cut (inside the first frame,
    mid-sequence, last sample, nothing), including the piecewise (chunked) schedule."""
    from fq3hip.codec import HipSpeechTokenizer
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, dtype, parts=("codec",))
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", dtype, max_frames=64)
    g = torch.Generator().manual_seed(3)
    for T in (3, 33, 64):
        codes = torch.randint(0, cfg.codec.codebook_size, (T, cfg.codec.num_quantizers), generator=g).cuda()
        full = tok.decode_tensor(codes)
        n = full.numel()
        assert n == tok.num_samples_total(T)
        for first in (1, 7, n // 3, n // 2 + 11, n - 1921, n - 1, n):
            if first < 0:
                continue
            tail = tok.decode_tensor(codes, first)
            assert tail.numel() == n - first and torch.equal(tail, full[first:]), (T, first)
    tok.CHUNK_FRAMES = 39
    codes = torch.randint(0, cfg.codec.codebook_size, (100, cfg.codec.num_quantizers), generator=g).cuda()
    full = tok.decode_tensor(codes)
    assert full.numel() == tok.num_samples_total(100)
    for first in (5, 39 * 1920 - 3, 39 * 1920 + 3, full.numel() - 4000):
        assert torch.equal(tok.decode_tensor(codes, first), full[first:]), first


def test_codec_tail_decode_is_bit_identical_real_shapes():
    """The streaming cuts at the benchmark's shapes: phase 1 (170 reference + 8..32 generated frames, keep the last 8
    frames' samples) and phase 2 (25 context + 8 new frames)."""
    cfg, tok = _real_codec(torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    codes = torch.randint(0, cfg.codec.codebook_size, (202, cfg.codec.num_quantizers), generator=g).cuda()
    for T, keep_frames in ((178, 8), (202, 8), (33, 8), (33, 33)):
        c = codes[:T].contiguous()
        full = tok.decode_tensor(c)
        first = max(0, full.numel() - keep_frames * 1920 - 37)
        tail = tok.decode_tensor(c, first)
        assert torch.equal(tail, full[first:]), (T, keep_frames)
    tok.close()


# ---- reference-prefix states (fq3_codec_prefix_*, round 6) -------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["bf16", "bf16x2", "fp32"])
def test_codec_prefix_state_is_bit_identical_real_shapes(precision):
    """The ICL call sites decode `reference codes + generated codes` (model.py:919-937; every phase-1 streaming chunk again,
    :1085-1115).  A prefix state (``tok.prefix_for(ref)``: the pre_conv context rows, every layer's post-RoPE K / V of the last 71
    reference rows, the last 48 output rows) lets the front end run over the rows BEHIND the reference only.  At the benchmark's
    shapes -- 170 reference frames + 8 .. 32 generated frames (phase 1, the last 8 frames' samples kept), + 130 / 200 generated
    frames with the reference's share cut off (non-streaming; 370 frames = two pieces) -- the waveform must equal the full decode's
    bit for bit, single and batched (different voices in one call), in all three precisions; a cut that reaches further back than the
    cached rows silently takes the full path; a reference shorter than the attention window / the cached output rows works too."""
    from fq3hip.codec import HipSpeechTokenizer
    from fq3hip.config import qwen3_tts_0p6b
    cfg = qwen3_tts_0p6b()
    dt = torch.float32 if precision == "fp32" else torch.bfloat16
    W = synth_weights(cfg, 0, dt, parts=("codec",), codec_normalized=True)
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", max_frames=400, precision=precision)
    g = torch.Generator().manual_seed(11)
    nq, cb = cfg.codec.num_quantizers, cfg.codec.codebook_size
    refs = [torch.randint(0, cb, (170, nq), generator=g).cuda() for _ in range(3)]
    gens = [torch.randint(0, cb, (200, nq), generator=g).cuda() for _ in range(3)]
    tok.use_prefix = True
    pfs = [tok.prefix_for(r) for r in refs]
    assert all(p is not None and p.ref_len == 170 for p in pfs)
    assert tok.prefix_for(refs[0]) is pfs[0]                                   # cached by identity
    assert tok.prefix_for(refs[0].cpu()) is tok.prefix_for(refs[0].cpu().clone())       # host tensors: by content
    for n_gen, keep in ((8, 8), (16, 8), (32, 8), (130, None), (200, None)):
        T = 170 + n_gen
        fulls = [torch.cat([r, gcodes[:n_gen]]) for r, gcodes in zip(refs, gens)]
        n = tok.num_samples_total(T)
        first = n - keep * 1920 - 37 if keep else int(170 / T * n)
        want = [tok.decode_tensor(f, first) for f in fulls]
        got = [tok.decode_tensor(f, first, prefix=p) for f, p in zip(fulls, pfs)]
        for i, (a, b) in enumerate(zip(want, got)):
            assert a.shape == b.shape and torch.equal(a, b), (precision, n_gen, "single", i)
        gotb = tok.decode_tensor_batch(torch.stack(fulls), first, prefixes=pfs)
        for i in range(3):
            assert torch.equal(gotb[i], want[i]), (precision, n_gen, "batched", i)
    # samples of the reference part itself: further back than the cached rows -> the full path, same values
    f = torch.cat([refs[0], gens[0][:8]])
    assert torch.equal(tok.decode_tensor(f, 100 * 1920, prefix=pfs[0]), tok.decode_tensor(f, 100 * 1920))
    assert torch.equal(tok.decode_tensor(f, 0, prefix=pfs[0]), tok.decode_tensor(f, 0))
    # short references: fewer rows than the attention window (71) / than the cached output rows (48) / than the pre_conv context (2)
    for rl in (1, 2, 30, 60):
        r = torch.randint(0, cb, (rl, nq), generator=g).cuda()
        f = torch.cat([r, gens[1][:24]])
        p = tok.prefix_for(r)
        n = tok.num_samples_total(rl + 24)
        for first in (int(rl / (rl + 24) * n), n - 8 * 1920):
            assert torch.equal(tok.decode_tensor(f, first, prefix=p), tok.decode_tensor(f, first)), (rl, first)
    tok.close()


def test_codec_prefix_state_tiny_decoder_and_streaming_vocoder():
    """The same property on the tiny test decoder (head_dim 32, window 8, other channel counts), and end to end through the
    streaming windowing state machine: the chunks StreamingVocoder produces with the tokenizer's prefix states switched on are those it
    produces with them switched off, bit for bit (phase 1 behind the reference, phase 2 without it)."""
    from fq3hip.codec import HipSpeechTokenizer
    from fq3hip.model import StreamingVocoder
    from fq3hip.streams import concurrent_stream
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("codec",))
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", max_frames=128)
    g = torch.Generator().manual_seed(3)
    nq, cb = cfg.codec.num_quantizers, cfg.codec.codebook_size
    ref = torch.randint(0, cb, (21, nq), generator=g).cuda()
    gen = torch.randint(0, cb, (56, nq), generator=g).cuda()
    side = concurrent_stream(torch.device("cuda"))
    outs = []
    for use in (False, True):
        tok.use_prefix = use
        voc = StreamingVocoder(tok, ref, 8, "cuda", side)
        assert (voc.prefix is not None) == use
        chunks = []
        for i in range(0, 56, 8):
            a, _sr = voc.push(gen[i:i + 8].contiguous())
            chunks.append(np.asarray(a).copy())
        outs.append(chunks)
    assert len(outs[0]) == len(outs[1]) == 7
    for i, (a, b) in enumerate(zip(*outs)):
        assert a.shape == b.shape and np.array_equal(a, b), i
    tok.close()


# ---- the bf16 x 2 high-precision mode (FQ3_BF16X2) and the batched decode (fq3_codec_decode_batch) --------------------------
@pytest.mark.parametrize("T", [40, 100])
def test_codec_bf16x2_mode_meets_1e_3_at_real_shapes(T, golden_dir):
    """codec_precision="bf16x2" on a bf16 weight table: weights stay bf16, every activation is a bf16 high part + a bf16 residual
    (16 mantissa bits), every product runs as two bf16 MFMAs.  Against the fp32-ARITHMETIC oracle on the same bf16-valued weights
    (tests/golden/codec_real_q.npz) the waveform must be within the north star's 1e-3 PCM RMS with a wide margin (gate 2e-4); a tail
    decode stays bit-identical to the tail of the full decode (the property the streaming call sites rely on)."""
    from fq3hip.codec import HipSpeechTokenizer
    from fq3hip.config import qwen3_tts_0p6b
    g = np.load(os.path.join(golden_dir, "codec_real_q.npz"))
    cfg = qwen3_tts_0p6b()
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("codec",), codec_normalized=True)          # the "checkpoint": bf16
    codes = torch.from_numpy(g[f"codes_{T}"].astype(np.int64)).cuda()
    ref = g[f"pcm_f32q_{T}"]
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", max_frames=208, precision="bf16x2")
    full = tok.decode_tensor(codes)
    wav = full.cpu().numpy()
    err = _rms(wav - ref)
    print(f"[parity] codec T={T} bf16x2 mode vs fp32-arithmetic oracle on the bf16 checkpoint weights: {err:.3e} (signal RMS {_rms(ref):.3f})")
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"parity_codec_bf16x2_T{T}.txt"), "w") as f:
        f.write(f"bf16x2_mode_vs_f32q {err:.4e}\n")
    assert wav.shape == ref.shape and err <= 2e-4, err
    n = full.numel()
    for first in (n - 8 * 1920, n // 2 + 7):
        assert torch.equal(tok.decode_tensor(codes, first), full[first:]), first
    tok.close()


def test_codec_bf16x2_tiny_matches_fp32_oracle():
    """The same mode on the tiny test decoder (other channel counts, head_dim 32, all tile shapes of small problems) against the fp32
    oracle run on the bf16-valued weights."""
    from fq3hip.codec import HipSpeechTokenizer
    from oracle import qwen3tts_oracle as O
    cfg = tiny_test_config()
    Wb = synth_weights(cfg, 0, torch.bfloat16, parts=("codec",))
    Wf = {k: v.float() for k, v in Wb.items()}
    tok = HipSpeechTokenizer(cfg.codec, Wb, "cuda", max_frames=64, precision="bf16x2")
    g = torch.Generator().manual_seed(17)
    for T in (1, 7, 40):
        codes = torch.randint(0, cfg.codec.codebook_size, (T, cfg.codec.num_quantizers), generator=g)
        ref = O.codec_decode(codes, Wf, cfg.codec).numpy()
        wav = tok.decode_tensor(codes.cuda()).cpu().numpy()
        assert wav.shape == ref.shape
        err, sig = _rms(wav - ref), _rms(ref)
        print(f"[parity] tiny codec bf16x2 T={T}: RMS {err:.3e} (signal {sig:.3f})")
        assert err <= 1e-3 * max(1.0, sig / 0.17), (T, err, sig)
    tok.close()


@pytest.mark.parametrize("precision", ["bf16", "fp32", "bf16x2"])
def test_codec_batched_decode_is_bit_identical_per_utterance_tiny(precision):
    """fq3_codec_decode_batch: B utterances through one set of launches == B single decodes, bit for bit -- full decodes, tail
    decodes, the piecewise schedule, and the [B, T, 16] payload of the reference's vocoder interface (model.py:924)."""
    from fq3hip.codec import HipSpeechTokenizer
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("codec",))
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", max_frames=64, precision=precision)
    g = torch.Generator().manual_seed(23)
    for B, T in ((3, 33), (5, 7), (2, 64)):
        codes = torch.randint(0, cfg.codec.codebook_size, (B, T, cfg.codec.num_quantizers), generator=g).cuda()
        single = [tok.decode_tensor(codes[b]) for b in range(B)]
        batch = tok.decode_tensor_batch(codes)
        assert batch.shape == (B, single[0].numel())
        for b in range(B):
            assert torch.equal(batch[b], single[b]), (B, T, b)
        n = single[0].numel()
        for first in (n // 2 + 3, max(0, n - 1921)):
            tb = tok.decode_tensor_batch(codes, first)
            for b in range(B):
                assert torch.equal(tb[b], single[b][first:]), (B, T, b, first)
        wavs, sr = tok.decode({"audio_codes": codes})
        assert sr == 24000 and len(wavs) == B and all(torch.equal(w, s) for w, s in zip(wavs, single))
    tok.CHUNK_FRAMES = 39                                         # pieces of 39 new + 25 context frames, batched piece by piece
    codes = torch.randint(0, cfg.codec.codebook_size, (3, 100, cfg.codec.num_quantizers), generator=g).cuda()
    batch = tok.decode_tensor_batch(codes)
    for b in range(3):
        assert torch.equal(batch[b], tok.decode_tensor(codes[b])), b
    # utterances of different lengths: pad to the longest, keep each one's own sample count (the decoder is causal)
    tok.CHUNK_FRAMES = 300
    padded = codes[:, :64].clone()
    padded[1, 61:] = 0                                            # utterance 1 has 61 frames; its tail is padding
    got = tok.decode_tensor_batch(padded)[1]
    assert torch.equal(got[: tok.num_samples_total(61)], tok.decode_tensor(padded[1, :61].contiguous()))
    tok.close()


@pytest.mark.parametrize("precision", ["bf16", "bf16x2"])
def test_codec_batched_decode_real_shapes(precision):
    """The benchmark's shapes: 4 utterances of 100 frames (> the attention window; big 256-wide tiles, glds tiles, the fused residual
    units all take their batched forms) and the first streaming chunk of 6 lanes (178 frames in, the last 8 frames' samples out)."""
    from fq3hip.codec import HipSpeechTokenizer
    from fq3hip.config import qwen3_tts_0p6b
    cfg = qwen3_tts_0p6b()
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("codec",), codec_normalized=True)
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", max_frames=208, precision=precision)
    g = torch.Generator().manual_seed(29)
    codes = torch.randint(0, cfg.codec.codebook_size, (4, 100, cfg.codec.num_quantizers), generator=g).cuda()
    batch = tok.decode_tensor_batch(codes)
    for b in range(4):
        assert torch.equal(batch[b], tok.decode_tensor(codes[b])), b
    codes = torch.randint(0, cfg.codec.codebook_size, (6, 178, cfg.codec.num_quantizers), generator=g).cuda()
    first = tok.num_samples_total(178) - 8 * 1920
    tail = tok.decode_tensor_batch(codes, first)
    for b in range(6):
        assert torch.equal(tail[b], tok.decode_tensor(codes[b], first)), b
    tok.close()
