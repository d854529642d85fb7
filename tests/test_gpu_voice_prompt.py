"""GPU: ``generate_voice_clone(ref_audio=...)`` end to end on the HIP analysers (tiny model): the prompt the wrapper builds
from a WAV equals the one built from the analysers' outputs handed over explicitly; the write-through voice cache lets a
model WITHOUT analyser weights serve the same reference afterwards."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

from fq3hip.config import tiny_test_config                                   # noqa: E402
from fq3hip.weights import synth_ref_audio_weights, synth_weights            # noqa: E402
from tests.test_refenc_oracle_pins import make_wave                          # noqa: E402

KW = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0, repetition_penalty=1.0, min_new_tokens=0, max_new_tokens=10)


def _model(with_analysers: bool):
    from fq3hip.model import FasterQwen3TTS
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("talker", "predictor", "codec", "text"))
    if with_analysers:
        W.update(synth_ref_audio_weights(cfg.ref_audio, 4))
    m = FasterQwen3TTS.from_weights(cfg, W, device="cuda", dtype=torch.float32, max_seq_len=512, codec_max_frames=400, max_frames=64)
    m.predictor_graph.do_sample = False
    m.predictor_graph.top_k = 0
    return cfg, W, m


def test_ref_audio_analysed_on_the_gpu_and_cached(tmp_path):
    from fq3hip import audio_io
    from oracle import refenc_oracle as RO
    cfg, W, m = _model(True)
    wav = str(tmp_path / "ref.wav")
    x = make_wave(48 * 30, seed=21).numpy()                    # 30 reference frames at the tiny encoder's 48 samples per frame
    audio_io.write_wav(wav, x, 24000)
    stored, _ = audio_io.read_wav(wav)                         # 16-bit quantised: what the analysers really see
    # ICL mode (the default): 0.5 s of silence is appended before analysis (model.py:443)
    with_sil = np.concatenate([stored, np.zeros(12000, np.float32)])
    items = m.model.create_voice_clone_prompt(ref_audio=(with_sil, 24000), ref_text="the reference")
    it = items[0]
    assert it.icl_mode and not it.x_vector_only_mode and it.ref_text == "the reference"
    assert it.ref_code.shape == (RO.encoded_length(cfg.ref_audio, len(with_sil)), 16) and it.ref_code.dtype == torch.long
    assert int(it.ref_code.max()) < cfg.ref_audio.codebook_size and it.ref_spk_embedding.shape == (cfg.talker.hidden_size,)
    ref_codes, margins = RO.tokenizer_encode(W, cfg.ref_audio, torch.from_numpy(with_sil), return_all=True)[:2]
    agree = (it.ref_code.cpu() == ref_codes)
    assert agree[:, 0].float().mean() > 0.9                    # silence frames sit on near-ties; the first level is stable
    spk_ref, _ = RO.speaker_embedding(W, cfg.ref_audio, torch.from_numpy(with_sil))
    assert torch.allclose(it.ref_spk_embedding.float().cpu(), spk_ref, atol=2e-3 * float(spk_ref.abs().max()))

    m.set_voice_ref_cache(str(tmp_path / "voices"))
    a, sr = m.generate_voice_clone(text="Hello there.", language="English", ref_audio=wav, ref_text="the reference", **KW)
    vcp = m.model._prompt_items_to_voice_clone_prompt(items)
    b, _ = m.generate_voice_clone(text="Hello there.", language="English", ref_text="the reference", voice_clone_prompt=vcp, **KW)
    assert sr == 24000 and len(a[0]) > 1000 and np.array_equal(a[0], b[0])
    # x-vector-only mode: no silence, no codes
    c, _ = m.generate_voice_clone(text="Hello there.", language="English", ref_audio=wav, xvec_only=True, **KW)
    xi = m.model.create_voice_clone_prompt(ref_audio=wav, x_vector_only_mode=True)[0]
    assert xi.ref_code is None and not xi.icl_mode
    d, _ = m.generate_voice_clone(text="Hello there.", language="English",
                                  voice_clone_prompt=dict(ref_code=[None], ref_spk_embedding=[xi.ref_spk_embedding],
                                                          x_vector_only_mode=[True], icl_mode=[False]), **KW)
    assert np.array_equal(c[0], d[0]) and not np.array_equal(c[0][:1000], a[0][:1000])
    # both analyses were written through: a model without analyser weights now serves the same WAV from disk
    files = sorted(os.listdir(tmp_path / "voices"))
    assert len([f for f in files if f.endswith(".spk")]) == 2 and len([f for f in files if f.endswith(".rvq")]) == 1
    _, _, m2 = _model(False)
    with pytest.raises(NotImplementedError, match="voice-reference cache"):
        m2.generate_voice_clone(text="Hello there.", language="English", ref_audio=wav, ref_text="the reference", **KW)
    m2.set_voice_ref_cache(str(tmp_path / "voices"))
    e, _ = m2.generate_voice_clone(text="Hello there.", language="English", ref_audio=wav, ref_text="the reference", **KW)
    # the cache stores the x-vector as float32 and the ids exactly: same prompt, same audio
    assert np.array_equal(e[0], a[0])


def test_speech_tokenizer_encode_duck_type_and_errors(tmp_path):
    cfg, W, m = _model(True)
    tok = m.model.model.speech_tokenizer
    x = make_wave(48 * 12, seed=2).numpy()
    out = tok.encode(x, sr=24000)
    assert len(out.audio_codes) == 1 and out.audio_codes[0].shape == (12, 16)
    out2 = tok.encode([x, x[: 48 * 5]], sr=24000)
    assert [c.shape[0] for c in out2.audio_codes] == [12, 5] and torch.equal(out2.audio_codes[0], out.audio_codes[0])
    # a 48 kHz clip is resampled on the host before analysis
    x48 = np.repeat(x, 2)
    assert tok.encode(x48, sr=48000).audio_codes[0].shape == (12, 16)
    with pytest.raises(ValueError, match="ref_text is required"):
        m.model.create_voice_clone_prompt(ref_audio=(x, 24000), ref_text="")
    with pytest.raises(ValueError, match="ref_audio is required"):
        m.model.create_voice_clone_prompt(ref_audio=None, ref_text="x")


def test_from_pretrained_checkpoint_directory_serves_ref_audio(tmp_path):
    """The reference's primary call sequence -- from_pretrained(dir) then generate_voice_clone(text, language,
    ref_audio=wav, ref_text=...) -- over a checkpoint directory in the (recalled) upstream layout."""
    import os
    from fq3hip import audio_io
    from fq3hip.model import FasterQwen3TTS
    from tests.test_loader import _write_checkpoint
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("talker", "predictor", "codec", "text"))
    W.update(synth_ref_audio_weights(cfg.ref_audio, 4))
    root = str(tmp_path / "ckpt")
    os.makedirs(root)
    _write_checkpoint(root, cfg, W)
    m = FasterQwen3TTS.from_pretrained(root, device="cuda", dtype=torch.float32, max_seq_len=512)
    assert m.model.ref_analyzer is not None and m.model.ref_analyzer.has_encoder and m.model.ref_analyzer.has_speaker
    m.predictor_graph.do_sample = False
    m.predictor_graph.top_k = 0
    wav = str(tmp_path / "ref.wav")
    audio_io.write_wav(wav, make_wave(48 * 20, seed=8).numpy(), 24000)
    a, sr = m.generate_voice_clone(text="Hello there.", language="English", ref_audio=wav, ref_text="the reference", **KW)
    # same answer as the in-memory model of the other tests given the same weights
    _, _, m2 = _model(True)
    b, _ = m2.generate_voice_clone(text="Hello there.", language="English", ref_audio=wav, ref_text="the reference", **KW)
    assert sr == 24000 and len(a[0]) > 1000 and np.array_equal(a[0], b[0])
