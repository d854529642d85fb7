"""HIP reference-audio analysers (fq3_refenc_*) vs the pinned CPU restatement (oracle/refenc_oracle.py)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

pytestmark = pytest.mark.gpu

from fq3hip.config import RefAudioConfig, tiny_ref_audio_config            # noqa: E402
from fq3hip.weights import synth_ref_audio_weights                          # noqa: E402
from tests.test_refenc_oracle_pins import make_wave                         # noqa: E402


def _compare_codes(rc, got, ref, margins, tie=1e-3):
    """Identical ids up to the first near-tie of each (frame, quantizer group) chain; returns (#decisions compared,
    #chains that diverged at a near-tie)."""
    assert got.shape == ref.shape
    n_cmp = n_div = 0
    for t in range(ref.shape[0]):
        for part in (range(0, rc.num_semantic_quantizers), range(rc.num_semantic_quantizers, rc.num_quantizers)):
            for lv in part:
                n_cmp += 1
                if got[t, lv] != ref[t, lv]:
                    assert margins[t, lv] < tie, f"frame {t} level {lv}: ids {int(got[t, lv])} vs {int(ref[t, lv])}, oracle margin {float(margins[t, lv]):.3e}"
                    n_div += 1
                    break
    return n_cmp, n_div


@pytest.mark.parametrize("n", [48 * 40, 48 * 40 + 1, 48 * 13 + 29, 97, 5000])
def test_tokenizer_encoder_tiny_shapes(n):
    from fq3hip.refenc import HipRefAudioAnalyzer
    from oracle import refenc_oracle as RO
    rc = tiny_ref_audio_config()
    W = synth_ref_audio_weights(rc, 0)
    an = HipRefAudioAnalyzer(rc, W)
    x = make_wave(n, seed=n)
    ref, margins = RO.tokenizer_encode(W, rc, x, return_all=True)[:2]
    got = an.encode(x).cpu()
    assert got.shape[0] == RO.encoded_length(rc, n) == an.num_frames(n)
    n_cmp, n_div = _compare_codes(rc, got, ref, margins)
    assert n_div <= max(1, ref.shape[0] // 10), (n_cmp, n_div)
    # a second call on the same object (workspace reuse) is bit-identical
    assert torch.equal(an.encode(x).cpu(), got)


def test_tokenizer_encoder_real_shapes():
    """Mimi-default shapes (64..1024 channels, strides 4 5 6 8, 8 x 512-d layers, window 250, 16 x 2048 codes), 3.1 s:
    long enough for 39 frames, > 64 keys per query in the 25 Hz transformer."""
    from fq3hip.refenc import HipRefAudioAnalyzer
    from oracle import refenc_oracle as RO
    rc = RefAudioConfig()
    W = synth_ref_audio_weights(rc, 1)
    an = HipRefAudioAnalyzer(rc, W)
    x = make_wave(24000 * 3 + 2500, seed=7)
    ref, margins = RO.tokenizer_encode(W, rc, x, return_all=True)[:2]
    got = an.encode(x).cpu()
    assert got.shape == ref.shape == (39, 16)
    n_cmp, n_div = _compare_codes(rc, got, ref, margins)
    exact = int((got == ref).sum())
    print(f"real-shape encoder: {exact}/{ref.numel()} ids identical, {n_div} chains diverged at near-ties")
    assert n_div <= 2 and exact >= ref.numel() - 40


@pytest.mark.parametrize("tiny", [True, False])
def test_speaker_encoder(tiny):
    from fq3hip.refenc import HipRefAudioAnalyzer
    from oracle import refenc_oracle as RO
    rc = tiny_ref_audio_config() if tiny else RefAudioConfig()
    W = synth_ref_audio_weights(rc, 2)
    an = HipRefAudioAnalyzer(rc, W)
    x = make_wave(7000 if tiny else 24000 * 4 + 123, seed=11)
    ref, mel_ref = RO.speaker_embedding(W, rc, x)
    emb, mel = an.speaker_embedding(x, return_mel=True)
    assert mel.shape == mel_ref.shape
    # log of near-silent bins amplifies fp32 DFT rounding; compare where the mel energy is not clamped, and overall RMS
    d = (mel.cpu() - mel_ref)
    assert d.abs().max() < 5e-3 and d.pow(2).mean().sqrt() < 2e-4, (float(d.abs().max()), float(d.pow(2).mean().sqrt()))
    e = (emb.cpu() - ref)
    scale = float(ref.abs().max())
    print(f"speaker embedding: max |diff| {float(e.abs().max()):.3e} on scale {scale:.3f}")
    assert e.abs().max() < 2e-3 * scale
    assert torch.equal(an.speaker_embedding(x).cpu(), emb.cpu())


def test_errors_and_partial_binding():
    from fq3hip import _lib as L
    from fq3hip.refenc import HipRefAudioAnalyzer
    rc = tiny_ref_audio_config()
    W = synth_ref_audio_weights(rc, 0)
    only_spk = {k: v for k, v in W.items() if k.startswith("speaker_encoder.")}
    an = HipRefAudioAnalyzer(rc, only_spk)
    assert an.has_speaker and not an.has_encoder
    with pytest.raises(L.Fq3Error, match="encoder weights not finalized"):
        an.encode(make_wave(2000))
    with pytest.raises(L.Fq3Error, match="too short"):
        an.speaker_embedding(make_wave(40))
    with pytest.raises(L.Fq3Error, match="no encoder"):
        HipRefAudioAnalyzer(rc, {})
    broken = dict(W)
    del broken["encoder.encoder.layers.4.block.1.conv.weight"]
    with pytest.raises(KeyError):
        HipRefAudioAnalyzer(rc, broken)


def test_against_transformers_module_outputs(golden_dir):
    """tests/golden/refenc.npz holds what the transformers modules themselves (MimiModel.encode, ECAPA_TimeDelayNet) produced for
    these seeds on CPU (oracle/make_golden_refenc.py): the HIP kernels against the modules, not against the restatement."""
    import numpy as np
    from fq3hip.refenc import HipRefAudioAnalyzer
    g = np.load(os.path.join(golden_dir, "refenc.npz"))
    rc = RefAudioConfig()
    wseed, n, seed = (int(v) for v in g["enc_meta"])
    an = HipRefAudioAnalyzer(rc, synth_ref_audio_weights(rc, wseed))
    got = an.encode(make_wave(n, seed=seed)).cpu()
    ref, margins = torch.from_numpy(g["enc_codes_mimi"].astype(np.int64)), torch.from_numpy(g["enc_margins"])
    n_cmp, n_div = _compare_codes(rc, got, ref, margins)
    print(f"vs MimiModel.encode: {int((got == ref).sum())}/{ref.numel()} ids identical, {n_div} chains diverged at near-ties")
    assert n_div <= 2 and int((got == ref).sum()) >= ref.numel() - 40
    wseed, n, seed = (int(v) for v in g["spk_meta"])
    an2 = HipRefAudioAnalyzer(rc, synth_ref_audio_weights(rc, wseed))
    emb, mel = an2.speaker_embedding(make_wave(n, seed=seed), return_mel=True)
    want = torch.from_numpy(g["spk_xvector_ecapa"])
    assert (emb.cpu() - want).abs().max() < 2e-3 * float(want.abs().max())
    assert (mel[:8].cpu() - torch.from_numpy(g["spk_mel_first_frames"])).abs().max() < 5e-3
