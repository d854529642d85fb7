"""Pins oracle/refenc_oracle.py (the restatement the GPU parity tests trust) against the transformers modules it was
restated from, instantiated with the same seeded weights: MimiModel.encode for the speech-tokenizer encoder,
ECAPA_TimeDelayNet for the speaker encoder, and torch.stft for the mel front end's DFT table."""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

from fq3hip.config import RefAudioConfig, tiny_ref_audio_config            # noqa: E402
from fq3hip.weights import synth_ref_audio_weights                          # noqa: E402
from oracle import refenc_oracle as RO                                      # noqa: E402


def make_wave(n, seed=0, sr=24000):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n, dtype=torch.float64) / sr
    x = 0.2 * torch.sin(2 * math.pi * 220 * t) + 0.1 * torch.sin(2 * math.pi * 1330 * t + 0.4) * torch.sin(2 * math.pi * 3 * t)
    return (x + 0.05 * torch.randn(n, generator=g, dtype=torch.float64)).float()


def _mimi(rc, W):
    from transformers import MimiConfig, MimiModel
    cfg = MimiConfig(num_filters=rc.num_filters, upsampling_ratios=list(reversed(rc.ratios)), kernel_size=rc.kernel_size,
                     last_kernel_size=rc.last_kernel_size, residual_kernel_size=rc.residual_kernel_size,
                     num_residual_layers=rc.num_residual_layers, dilation_growth_rate=rc.dilation_growth_rate, compress=rc.compress,
                     hidden_size=rc.hidden_size, num_hidden_layers=rc.num_hidden_layers, num_attention_heads=rc.num_attention_heads,
                     num_key_value_heads=rc.num_attention_heads, head_dim=rc.head_dim, intermediate_size=rc.intermediate_size,
                     sliding_window=rc.sliding_window, norm_eps=rc.norm_eps, num_quantizers=rc.num_quantizers,
                     num_semantic_quantizers=rc.num_semantic_quantizers, codebook_size=rc.codebook_size, codebook_dim=rc.codebook_dim,
                     vector_quantization_hidden_dimension=rc.codebook_dim, upsample_groups=rc.hidden_size,
                     max_position_embeddings=rc.max_positions, attn_implementation="eager")
    m = MimiModel(cfg).eval()
    sd = {k[len("encoder."):]: v for k, v in W.items() if k.startswith("encoder.")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith(("decoder", "upsample", "quantizer.semantic_residual_vector_quantizer.output_proj",
                             "quantizer.acoustic_residual_vector_quantizer.output_proj")) or k.endswith("initialized")
               for k in missing), [k for k in missing][:8]
    return m


@pytest.mark.parametrize("n", [48 * 20, 48 * 20 + 17, 1111])
def test_tokenizer_encoder_restatement_matches_mimi(n):
    rc = tiny_ref_audio_config()
    W = synth_ref_audio_weights(rc, 0)
    m = _mimi(rc, W)
    x = make_wave(n, seed=n)
    with torch.no_grad():
        ref = m.encode(x.reshape(1, 1, -1), num_quantizers=rc.num_quantizers).audio_codes[0].transpose(0, 1)
        emb_ref = m.encoder_transformer(m.encoder(x.reshape(1, 1, -1)).transpose(1, 2))[0][0]
        codes, margins, h, d = RO.tokenizer_encode(W, rc, x, return_all=True)
    assert codes.shape == (RO.encoded_length(rc, n), rc.num_quantizers) == tuple(ref.shape)
    assert int(m.get_encoded_length(torch.tensor(n))) == codes.shape[0]
    assert torch.allclose(h, emb_ref, atol=2e-5, rtol=1e-4)
    # identical ids wherever the arg-min is not a near-tie (fp32 cdist vs an exact ordering); everything after the first
    # near-tie of a frame's chain is allowed to differ (the residuals diverge)
    bad = 0
    for t in range(codes.shape[0]):
        for part in (range(0, rc.num_semantic_quantizers), range(rc.num_semantic_quantizers, rc.num_quantizers)):
            for lv in part:
                if codes[t, lv] != ref[t, lv]:
                    assert margins[t, lv] < 1e-5, (t, lv, float(margins[t, lv]))
                    bad += 1
                    break
    assert bad <= max(1, codes.numel() // 100)


def test_tokenizer_encoder_restatement_real_shapes():
    """The real (Mimi-default) shapes once, on a quarter of a second of audio."""
    rc = RefAudioConfig()
    W = synth_ref_audio_weights(rc, 1)
    m = _mimi(rc, W)
    x = make_wave(6000, seed=5)
    with torch.no_grad():
        ref = m.encode(x.reshape(1, 1, -1), num_quantizers=rc.num_quantizers).audio_codes[0].transpose(0, 1)
        codes, margins = RO.tokenizer_encode(W, rc, x, return_all=True)[:2]
    assert codes.shape == ref.shape == (4, 16)
    agree = (codes == ref)
    assert agree[:, 0].all()
    for t in range(4):
        for lv in range(1, 16):
            if not agree[t, lv]:
                assert margins[t, lv] < 1e-5
                break


def _ecapa(rc, W):
    from transformers.models.qwen2_5_omni.configuration_qwen2_5_omni import Qwen2_5OmniDiTConfig
    from transformers.models.qwen2_5_omni.modeling_qwen2_5_omni import ECAPA_TimeDelayNet
    cfg = Qwen2_5OmniDiTConfig(mel_dim=rc.mel_dim, enc_dim=rc.enc_dim, enc_channels=list(rc.enc_channels),
                               enc_kernel_sizes=list(rc.enc_kernel_sizes), enc_dilations=list(rc.enc_dilations),
                               enc_attention_channels=rc.enc_attention_channels, enc_res2net_scale=rc.enc_res2net_scale,
                               enc_se_channels=rc.enc_se_channels)
    m = ECAPA_TimeDelayNet(cfg).eval()
    sd = {k[len("speaker_encoder."):]: v for k, v in W.items() if k.startswith("speaker_encoder.")}
    m.load_state_dict(sd, strict=True)
    return m


@pytest.mark.parametrize("tiny", [True, False])
def test_speaker_encoder_restatement_matches_ecapa(tiny):
    rc = tiny_ref_audio_config() if tiny else RefAudioConfig()
    W = synth_ref_audio_weights(rc, 2)
    m = _ecapa(rc, W)
    x = make_wave(9000 if tiny else 24000, seed=3)
    with torch.no_grad():
        emb, mel = RO.speaker_embedding(W, rc, x)
        ref = m(mel[None])[0]
    assert emb.shape == (rc.enc_dim,)
    assert torch.allclose(emb, ref, atol=1e-5, rtol=1e-4)
    assert emb.std() > 0.05                              # not a degenerate (saturated / constant) embedding


def test_mel_front_end_tables():
    """The product's DFT-as-GEMM and mel tables reproduce the torch.stft based front end of the oracle."""
    from fq3hip.refenc import dft_table, slaney_mel_basis
    rc = RefAudioConfig()
    x = make_wave(24000, seed=9)
    mel_ref = RO.mel_spectrogram(rc, x, torch.float64)
    pad = (rc.n_fft - rc.hop_size) // 2
    y = torch.nn.functional.pad(x.double()[None, None], (pad, pad), mode="reflect")[0, 0]
    frames = y.unfold(0, rc.n_fft, rc.hop_size)                                   # [F, n_fft]
    NB = rc.n_bins_padded
    spec = frames @ dft_table(rc.n_fft, NB).double().t()                          # [F, 2 NB]
    mag = torch.sqrt(spec[:, :NB] ** 2 + spec[:, NB:] ** 2 + 1e-9)
    basis = torch.zeros(rc.mel_dim, NB, dtype=torch.float64)
    basis[:, : rc.n_fft // 2 + 1] = torch.from_numpy(slaney_mel_basis(rc.sample_rate, rc.n_fft, rc.mel_dim, rc.fmin, rc.fmax)).double()
    mel = torch.log(torch.clamp(mag @ basis.t(), min=1e-5))
    assert mel.shape == mel_ref.shape == (24000 // rc.hop_size, rc.mel_dim)
    assert torch.allclose(mel, mel_ref, atol=2e-4, rtol=1e-4)
    # the two mel filterbank write-ups (product: vectorised, oracle: per filter) agree, and every filter is non-empty
    b2 = RO.slaney_mel(rc.sample_rate, rc.n_fft, rc.mel_dim, rc.fmin, rc.fmax)
    assert torch.allclose(basis[:, : rc.n_fft // 2 + 1], b2, atol=1e-7)
    assert (b2.sum(1) > 0).all()


def test_strided_conv_packing_is_a_two_tap_gemm_over_the_row_view():
    """Host logic of fq3hip/refenc.py: a causal conv with k = 2r, stride r equals a 2-tap stride-1 GEMM over the
    [T/r][r*C] view of the (zero-tailed) input with the packed weight [Cout][2][r*Cin] -- what csrc/fq3_refenc.hip runs."""
    from fq3hip.refenc import pack_ref_audio_weights
    rc = tiny_ref_audio_config()
    W = synth_ref_audio_weights(rc, 3)
    P = pack_ref_audio_weights(W, rc)
    g = torch.Generator().manual_seed(1)
    li, C = 1, rc.num_filters
    for r in rc.ratios:
        li += rc.num_residual_layers + 1
        name = f"encoder.encoder.layers.{li}.conv"
        for T in (5 * r, 5 * r + 1, 7 * r - 1):
            x = torch.randn(1, C, T, generator=g, dtype=torch.float64)
            ref = RO.mimi_conv1d(x, W[name + ".weight"].double(), W[name + ".bias"].double(), stride=r)[0].t()     # [T', 2C]
            Tn = -(-T // r)
            rows = torch.zeros(Tn * r, C, dtype=torch.float64)
            rows[:T] = x[0].t()
            view = rows.reshape(Tn, r * C)
            prev = torch.cat([torch.zeros(1, r * C, dtype=torch.float64), view[:-1]], 0)                            # tap offset -1 (causal)
            pw = P[name + ".weight"].double()                                                                        # [2C, 2, r*C]
            got = prev @ pw[:, 0].t() + view @ pw[:, 1].t() + W[name + ".bias"].double()
            assert got.shape == ref.shape == (Tn, 2 * C)
            assert torch.allclose(got, ref, atol=1e-10), (r, T)
        li += 1
        C *= 2
    # the stride-2 conv after the transformer: replicate padding = 2 copies of the first row, the last row repeated to even length
    H = rc.hidden_size
    for T in (6, 7):
        x = torch.randn(1, H, T, generator=g, dtype=torch.float64)
        ref = RO.mimi_conv1d(x, W["encoder.downsample.conv.weight"].double(), None, stride=2, pad_mode="replicate")[0].t()
        T5 = (T + 1) // 2
        idx = torch.clamp(torch.arange(2 + 2 * T5) - 2, 0, T - 1)
        padded = x[0].t()[idx].reshape(1 + T5, 2 * H)
        pw = P["encoder.downsample.conv.weight"].double()
        got = padded[:-1] @ pw[:, 0].t() + padded[1:] @ pw[:, 1].t()
        assert torch.allclose(got, ref, atol=1e-10), T


def test_refenc_golden_fixture_is_what_the_modules_produce(golden_dir):
    """tests/golden/refenc.npz (scored against by the GPU test) is reproduced here: the speaker half from the module itself,
    the encoder half through the restatement (MimiModel on a 3.1 s clip at the real shapes is the slow part of
    oracle/make_golden_refenc.py, which asserts their agreement when it writes the file)."""
    g = np.load(os.path.join(golden_dir, "refenc.npz"))
    rc = RefAudioConfig()
    wseed, n, seed = (int(v) for v in g["spk_meta"])
    W = synth_ref_audio_weights(rc, wseed)
    with torch.no_grad():
        emb, mel = RO.speaker_embedding(W, rc, make_wave(n, seed=seed))
        ref = _ecapa(rc, W)(mel[None])[0]
    assert np.allclose(ref.numpy(), g["spk_xvector_ecapa"], atol=1e-6) and np.allclose(mel[:8].numpy(), g["spk_mel_first_frames"], atol=1e-5)
    wseed, n, seed = (int(v) for v in g["enc_meta"])
    with torch.no_grad():
        codes, margins = RO.tokenizer_encode(synth_ref_audio_weights(rc, wseed), rc, make_wave(n, seed=seed), return_all=True)[:2]
    assert np.array_equal(codes.numpy(), g["enc_codes_mimi"].astype(np.int64))
    assert np.allclose(margins.numpy(), g["enc_margins"], rtol=1e-3, atol=1e-7)
