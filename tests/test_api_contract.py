"""CPU: public-surface contract of FasterQwen3TTS, mirroring what the reference pins in its own
tests/test_voice_clone_prompt_api.py (:57-97, :148-204, :259-386, :116-133) and tests/test_sample_rate.py."""
import inspect
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from fq3hip.model import FasterQwen3TTS


class _Graph:
    def __init__(self):
        self.calls = []

    def capture(self, **kw):
        self.calls.append(kw)


def _wrapper(base=None):
    base = base or SimpleNamespace(model=SimpleNamespace(speech_tokenizer=SimpleNamespace(sample_rate=24000)))
    return FasterQwen3TTS(base_model=base, predictor_graph=_Graph(), talker_graph=_Graph())


def test_signatures_match_reference_order_and_defaults():
    for name in ("generate_voice_clone", "generate_voice_clone_streaming"):
        sig = inspect.signature(getattr(FasterQwen3TTS, name))
        params = list(sig.parameters)
        assert params[:5] == ["self", "text", "language", "ref_audio", "ref_text"]
        assert params[5] == "max_new_tokens"
        assert params[-1] == "voice_clone_prompt"
        assert sig.parameters["xvec_only"].default is False
        assert sig.parameters["non_streaming_mode"].default is None
        assert sig.parameters["min_new_tokens"].default == 2
        assert sig.parameters["temperature"].default == 0.9
        assert sig.parameters["top_k"].default == 50
        assert sig.parameters["top_p"].default == 1.0
        assert sig.parameters["repetition_penalty"].default == 1.05
    s = inspect.signature(FasterQwen3TTS.generate_voice_clone_streaming)
    assert s.parameters["chunk_size"].default == 12 and s.parameters["parity_mode"].default is False
    fp = inspect.signature(FasterQwen3TTS.from_pretrained).parameters
    assert list(fp)[:6] == ["model_name", "device", "dtype", "attn_implementation", "max_seq_len", "backend"]
    assert fp["max_seq_len"].default == 2048 and fp["backend"].default == "torch" and fp["device"].default == "cuda"
    for name in ("generate_custom_voice", "generate_custom_voice_streaming"):
        assert list(inspect.signature(getattr(FasterQwen3TTS, name)).parameters)[:6] == \
            ["self", "text", "speaker", "language", "instruct", "non_streaming_mode"]
    for name in ("generate_voice_design", "generate_voice_design_streaming"):
        assert list(inspect.signature(getattr(FasterQwen3TTS, name)).parameters)[:5] == \
            ["self", "text", "instruct", "language", "non_streaming_mode"]


def test_warmup_is_idempotent_and_forwards_kwargs():
    w = _wrapper()
    w.warmup(prefill_len=42)
    w.warmup(prefill_len=7)
    w._warmup(9)
    assert w.predictor_graph.calls == [{"num_warmup": 3}]
    assert w.talker_graph.calls == [{"prefill_len": 42, "num_warmup": 3}]


def test_non_streaming_mode_defaults():
    assert FasterQwen3TTS._resolve_non_streaming_mode(None, default=False) is False
    assert FasterQwen3TTS._resolve_non_streaming_mode(None, default=True) is True
    assert FasterQwen3TTS._resolve_non_streaming_mode(True, default=False) is True
    assert FasterQwen3TTS._resolve_non_streaming_mode(False, default=True) is False


def test_ggml_only_arguments_are_rejected():
    w = _wrapper()
    for kw in ({"ref_spk": "a.spk"}, {"ref_rvq": "a.rvq"}, {"ref_spk_emb": np.zeros(4)}, {"ref_codes": np.zeros((2, 16))}):
        with pytest.raises(NotImplementedError, match="backend='ggml'"):
            w.generate_voice_clone("hi", "English", **kw)
        with pytest.raises(NotImplementedError, match="backend='ggml'"):
            next(w.generate_voice_clone_streaming("hi", "English", **kw))


def test_backend_and_device_validation():
    with pytest.raises(ValueError, match="Unsupported backend"):
        FasterQwen3TTS.from_pretrained("x", backend="onnx")
    with pytest.raises(ValueError, match="CUDA graphs require CUDA device"):
        FasterQwen3TTS.from_pretrained("x", device="cpu")
    with pytest.raises(NotImplementedError):
        w = _wrapper()
        w.generate("hello")


def _prompt_wrapper():
    base = SimpleNamespace(
        model=SimpleNamespace(speech_tokenizer=SimpleNamespace(sample_rate=24000)),
        _build_ref_text=lambda t: f"<ref>{t}",
        _tokenize_texts=lambda texts: [torch.tensor([[1, 2, 3, 9, 9, 4, 5]]) for _ in texts],
        _prompt_items_to_voice_clone_prompt=lambda items: dict(
            ref_code=[i.ref_code for i in items], ref_spk_embedding=[i.ref_spk_embedding for i in items],
            x_vector_only_mode=[i.x_vector_only_mode for i in items], icl_mode=[i.icl_mode for i in items]))
    return _wrapper(base)


def test_precomputed_prompt_validation():
    w = _prompt_wrapper()
    ids = [torch.zeros(1, 8, dtype=torch.long)]
    spk = torch.zeros(8)
    with pytest.raises(ValueError, match="missing required keys"):
        w._resolve_precomputed_voice_clone_prompt(ids, "", {"ref_code": [None]})
    with pytest.raises(ValueError, match="must be a list with length 1"):
        w._resolve_precomputed_voice_clone_prompt(ids, "", {"ref_spk_embedding": spk})
    with pytest.raises(ValueError, match="must be opposites"):
        w._resolve_precomputed_voice_clone_prompt(ids, "", {"ref_spk_embedding": [spk], "x_vector_only_mode": [True], "icl_mode": [True]})
    with pytest.raises(ValueError, match="ref_code must be None"):
        w._resolve_precomputed_voice_clone_prompt(ids, "", {"ref_spk_embedding": [spk], "x_vector_only_mode": [True], "ref_code": [torch.zeros(3, 16)]})
    with pytest.raises(ValueError, match="ref_code is required in ICL mode"):
        w._resolve_precomputed_voice_clone_prompt(ids, "t", {"ref_spk_embedding": [spk], "x_vector_only_mode": [False], "ref_code": [None]})
    with pytest.raises(ValueError, match="ref_text is required"):
        w._resolve_precomputed_voice_clone_prompt(ids, "", {"ref_spk_embedding": [spk], "x_vector_only_mode": [False], "ref_code": [torch.zeros(3, 16)]})
    vcp, ref_ids, icl = w._resolve_precomputed_voice_clone_prompt(ids, "", {"ref_spk_embedding": [spk]})
    assert icl is False and ref_ids == [None] and vcp["x_vector_only_mode"] == [True] and vcp["icl_mode"] == [False]
    vcp, ref_ids, icl = w._resolve_precomputed_voice_clone_prompt(
        ids, "hello", {"ref_spk_embedding": [spk], "x_vector_only_mode": [False], "ref_code": [torch.zeros(3, 16)]})
    assert icl is True and ref_ids[0] is not None
    with pytest.raises(ValueError, match="ref_audio is required"):
        w._resolve_voice_clone_prompt(ids, None, "", False, True, None)
    with pytest.raises(ValueError, match="must have length 1"):
        w._resolve_precomputed_voice_clone_prompt(ids, "", [])


def test_sample_rate_inference_order():
    a = SimpleNamespace(model=SimpleNamespace(speech_tokenizer=SimpleNamespace(sample_rate=16000)), sample_rate=8000)
    assert FasterQwen3TTS._infer_sample_rate(a) == 16000
    b = SimpleNamespace(model=SimpleNamespace(speech_tokenizer=None), sample_rate=8000)
    assert FasterQwen3TTS._infer_sample_rate(b) == 8000
    c = SimpleNamespace()
    assert FasterQwen3TTS._infer_sample_rate(c) == 24000
    w = _wrapper(b)
    with pytest.raises(AttributeError):
        _ = w.speech_tokenizer
    assert _wrapper().speech_tokenizer.sample_rate == 24000


def test_vocoder_payload_shape_and_empty_generation(monkeypatch):
    """decode payload is {"audio_codes": codes.unsqueeze(0)} (reference tests/test_sample_rate.py:53-75);
    an empty generation returns one zero sample (model.py:912-914)."""
    seen = {}

    class Tok:
        sample_rate = 24000

        def decode(self, payload):
            seen["shape"] = tuple(payload["audio_codes"].shape)
            return [torch.zeros(100)], 24000

    w = _wrapper(SimpleNamespace(model=SimpleNamespace(speech_tokenizer=Tok())))
    import fq3hip.generate as G
    codes = torch.zeros(7, 16, dtype=torch.long)
    monkeypatch.setattr(G, "fast_generate", lambda **kw: (codes, dict(steps=7, prefill_ms=1.0, decode_s=0.1, ms_per_step=1.0)))
    m = w.model.model
    audio, sr = w._run_full(m, None, None, None, None, None, None, None, {})
    assert seen["shape"] == (1, 7, 16) and sr == 24000 and audio[0].shape == (100,)
    ref = torch.zeros(3, 16, dtype=torch.long)
    audio, _ = w._run_full(m, None, None, None, None, None, None, ref, {})
    assert seen["shape"] == (1, 10, 16) and audio[0].shape == (70,)       # int(3/10*100) samples of reference cut
    monkeypatch.setattr(G, "fast_generate", lambda **kw: (None, dict(steps=0, prefill_ms=1.0, decode_s=0.1, ms_per_step=0)))
    audio, sr = w._run_full(m, None, None, None, None, None, None, None, {})
    assert len(audio) == 1 and audio[0].shape == (1,) and audio[0][0] == 0


def test_host_helpers():
    from fq3hip.native_model import ByteTokenizer
    from fq3hip.codec import pack_codec_weights
    from fq3hip.config import tiny_test_config, qwen3_tts_0p6b, qwen3_tts_1p7b, from_hf_config
    from fq3hip.weights import synth_weights
    ids = ByteTokenizer(512)("hey")
    assert ids[:3] == [1, 2, 3] and len(ids) == 3 + 3 + 5
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("codec",))
    P = pack_codec_weights(W, cfg.codec)
    c = cfg.codec
    assert P["decoder.pre_conv.conv.weight"].shape == (c.latent_dim, 3, c.codebook_dim)
    r = c.upsample_rates[0]
    assert P["decoder.decoder.1.block.1.conv.weight"].shape == (r * c.decoder_dim // 2, 2, c.decoder_dim)
    # transposed-conv packing: tap 0 holds kernel index q, tap 1 holds q + stride
    w = W["decoder.decoder.1.block.1.conv.weight"]
    q, co, ci = 3, 5, 7
    assert P["decoder.decoder.1.block.1.conv.weight"][q * (c.decoder_dim // 2) + co, 1, ci] == w[ci, co, q + r]
    # algorithmic bytes per frame = SURVEY.md section 8(d): 1109.4 MB + 114,688 B * p (0.6B), 3055.6 MB (1.7B)
    import bench
    assert abs(bench.algorithmic_bytes_per_frame(qwen3_tts_0p6b(), 0) / 1e6 - 1109.4) < 0.5
    assert abs(bench.algorithmic_bytes_per_frame(qwen3_tts_1p7b(), 0) / 1e6 - 3055.6) < 0.5
    assert bench.algorithmic_bytes_per_frame(qwen3_tts_0p6b(), 1) - bench.algorithmic_bytes_per_frame(qwen3_tts_0p6b(), 0) == 114688
    cfg2 = from_hf_config({"talker_config": {"hidden_size": 2048, "intermediate_size": 6144, "vocab_size": 3072,
                                             "code_predictor_config": {"hidden_size": 1024, "num_hidden_layers": 5}},
                           "tts_pad_token_id": 7})
    assert cfg2.talker.hidden_size == 2048 and cfg2.predictor_has_projection and cfg2.tts_pad_token_id == 7


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from fq3hip import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU / PyTorch fallback"):
        _lib.load()


def test_drop_in_import_name():
    """`from faster_qwen3_tts import FasterQwen3TTS` and the reference's submodule names resolve to fq3hip."""
    import importlib
    pkg = importlib.import_module("faster_qwen3_tts")
    from faster_qwen3_tts import FasterQwen3TTS as A
    assert A is FasterQwen3TTS
    for m in ("model", "generate", "streaming", "sampling", "talker_graph", "predictor_graph"):
        mod = importlib.import_module(f"faster_qwen3_tts.{m}")
        assert mod.__name__ == f"fq3hip.{m}"
    from faster_qwen3_tts.generate import fast_generate
    from faster_qwen3_tts.streaming import fast_generate_streaming, parity_generate_streaming
    from faster_qwen3_tts.sampling import sample_logits, apply_repetition_penalty
    import inspect
    # argument order of the decode loops = reference generate.py:16-37 / streaming.py:19-36
    assert list(inspect.signature(fast_generate).parameters)[:8] == [
        "talker", "talker_input_embeds", "attention_mask", "trailing_text_hiddens", "tts_pad_embed", "config",
        "predictor_graph", "talker_graph"]
    assert list(inspect.signature(fast_generate_streaming).parameters)[:8] == list(inspect.signature(fast_generate).parameters)[:8]
    assert inspect.signature(fast_generate_streaming).parameters["chunk_size"].default == 12
    assert [p for p in inspect.signature(sample_logits).parameters][:1] == ["logits"]
