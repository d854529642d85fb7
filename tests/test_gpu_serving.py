"""GPU: the serving shells over the real HIP path (tiny model): ref_audio served from the on-disk voice cache, the CLI
writing a WAV, and the OpenAI-compatible endpoint with the batch scheduler taking concurrent requests into lock-step lanes."""
import struct
import threading
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fq3hip.config import tiny_test_config
from fq3hip.weights import synth_weights


@pytest.fixture(scope="module")
def model():
    from fq3hip.model import FasterQwen3TTS
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("talker", "predictor", "codec", "text"))
    m = FasterQwen3TTS.from_weights(cfg, W, device="cuda", dtype=torch.float32, max_seq_len=160, codec_max_frames=128, max_frames=64)
    m._cfg_for_test = cfg
    return m


def _voice(cfg, n_ref=9):
    g = torch.Generator().manual_seed(12)
    spk = torch.randn(cfg.talker.hidden_size, generator=g)
    codes = torch.cat([torch.randint(0, cfg.talker.vocab_size - 1024, (n_ref, 1), generator=g),
                       torch.randint(0, cfg.codec.codebook_size, (n_ref, cfg.num_code_groups - 1), generator=g)], 1)
    return spk, codes


def test_ref_audio_is_served_from_the_voice_cache(model, tmp_path):
    from fq3hip import audio_io
    from fq3hip.voice_cache import VoiceRefCache, export_voice_clone_prompt
    cfg = model._cfg_for_test
    spk, codes = _voice(cfg)
    wav = str(tmp_path / "ref.wav")
    audio = (np.sin(np.arange(24000) / 30.0) * 0.3).astype(np.float32)
    audio_io.write_wav(wav, audio, 24000)
    kw = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0, repetition_penalty=1.0, min_new_tokens=0, max_new_tokens=10)
    model.predictor_graph.do_sample = False
    model.predictor_graph.top_k = 0
    with pytest.raises(NotImplementedError, match="voice-reference cache"):
        model.generate_voice_clone(text="Hello.", language="English", ref_audio=wav, ref_text="the reference", **kw)
    # export once (what a box with upstream qwen-tts would do), then the reference's primary call form works
    cache = VoiceRefCache(tmp_path / "voices")
    stored, sr = audio_io.read_wav(wav)
    with_silence = np.concatenate([stored, np.zeros(12000, np.float32)])
    ident = f"{model.model.model.tts_model_type}-{model.model.model.tts_model_size}"
    export_voice_clone_prompt(cache, with_silence, SimpleNamespace(ref_spk_embedding=spk, ref_code=codes, ref_text="the reference"),
                              append_silence=True, model_identity=ident)
    model.set_voice_ref_cache(str(tmp_path / "voices"))
    a, sr = model.generate_voice_clone(text="Hello.", language="English", ref_audio=wav, ref_text="the reference", **kw)
    vcp = dict(ref_code=[codes], ref_spk_embedding=[spk], x_vector_only_mode=[False], icl_mode=[True])
    b, _ = model.generate_voice_clone(text="Hello.", language="English", ref_text="the reference", voice_clone_prompt=vcp, **kw)
    assert sr == 24000 and np.array_equal(a[0], b[0]) and len(a[0]) > 1000
    model.set_voice_ref_cache(None)
    model._voice_prompt_cache.clear()


def test_cli_clone_writes_a_wav(model, tmp_path):
    from fq3hip import audio_io, cli
    cfg = model._cfg_for_test
    spk, codes = _voice(cfg)
    out = str(tmp_path / "cli.wav")
    args = cli.build_parser().parse_args(["clone", "--text", "A short line.", "--output", out, "--ref-audio", "unused.wav", "--ref-text", "r",
                                          "--max-new-tokens", "8", "--greedy", "--streaming", "--chunk-size", "4"])
    # the scripted call path: hand the prompt over directly (no reference-audio analysis on this path)
    orig = model.generate_voice_clone_streaming
    vcp = dict(ref_code=[codes], ref_spk_embedding=[spk], x_vector_only_mode=[False], icl_mode=[True])
    model.generate_voice_clone_streaming = lambda **kw: orig(**{**kw, "ref_audio": None, "voice_clone_prompt": vcp})
    try:
        cli.cmd_once(args, model=model)
    finally:
        model.generate_voice_clone_streaming = orig
    y, sr = audio_io.read_wav(out)
    assert sr == 24000 and len(y) > 1000 and np.abs(y).max() <= 1.0


def test_openai_endpoint_batch_scheduler_takes_concurrent_requests(model):
    from fastapi.testclient import TestClient
    from fq3hip.server import create_app
    cfg = model._cfg_for_test
    spk, codes = _voice(cfg)
    vcp = dict(ref_code=[codes], ref_spk_embedding=[spk], x_vector_only_mode=[False], icl_mode=[True])
    voices = {"alloy": {"voice_clone_prompt": vcp, "ref_text": "the reference", "language": "English", "max_new_tokens": 12}}
    client = TestClient(create_app(model, voices, default_voice="alloy", scheduler="batch", lanes=3))
    assert client.get("/health").json() == {"status": "ok", "model_loaded": True, "scheduler": "batch", "lanes": 3}
    texts = ["One.", "A second and longer request.", "Third.", "Number four goes last.", "Five."]
    results = [None] * len(texts)

    def call(i):
        results[i] = client.post("/v1/audio/speech", json={"input": texts[i], "voice": "alloy", "response_format": "wav"})

    th = [threading.Thread(target=call, args=(i,)) for i in range(len(texts))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    for r in results:
        assert r is not None and r.status_code == 200, None if r is None else r.text
        body = r.content
        n = struct.unpack("<I", body[40:44])[0]
        # the batch scheduler streams every utterance chunk by chunk: WAV header of unknown length, as the reference's server
        assert body[:4] == b"RIFF" and n == 0xFFFFFFFF and len(body) - 44 > 2000 and (len(body) - 44) % 2 == 0
    r = client.post("/v1/audio/speech", json={"input": "pcm please", "voice": "alloy", "response_format": "pcm"})
    assert r.status_code == 200 and len(r.content) > 2000 and len(r.content) % 2 == 0


def test_batch_streaming_equals_single_stream_streaming(model):
    """generate_voice_clone_batch_streaming: per utterance the same chunks (sizes and samples) as
    generate_voice_clone_streaming, while the utterances share lock-step lanes (greedy fp32: lanes are bit-identical to the
    single-stream decode)."""
    cfg = model._cfg_for_test
    spk, codes = _voice(cfg)
    vcp = dict(ref_code=[codes], ref_spk_embedding=[spk], x_vector_only_mode=[False], icl_mode=[True])
    kw = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0, repetition_penalty=1.0, min_new_tokens=0, ref_text="the reference",
              voice_clone_prompt=vcp, chunk_size=4)
    model.predictor_graph.do_sample = False
    model.predictor_graph.top_k = 0
    texts = ["First line.", "A second, rather longer line of text.", "Three.", "And the fourth one."]
    budgets = 23                                   # not a multiple of the chunk size: the last chunk is a remainder
    single = []
    for t in texts:
        single.append([(a.copy(), tm["is_final"]) for a, _sr, tm in model.generate_voice_clone_streaming(text=t, language="English",
                                                                                                      max_new_tokens=budgets, **kw)])
    got = {i: [] for i in range(len(texts))}
    order = []
    for i, a, sr, tm in model.generate_voice_clone_batch_streaming(texts, language="English", max_new_tokens=budgets, lanes=3, **kw):
        assert sr == 24000
        got[i].append((a, tm["is_final"], tm["chunk_index"]))
        order.append(i)
    assert len(set(order[:6])) == 3               # the first chunks of the three lanes arrive interleaved, not utterance by utterance
    for i in range(len(texts)):
        ref = [c for c in single[i] if len(c[0]) > 0]
        mine = [c for c in got[i] if len(c[0]) > 0]
        assert len(mine) == len(ref) and [c[2] for c in got[i]] == list(range(len(got[i])))
        for (a, _f), (b, _g, _k) in zip(ref, mine):
            assert a.shape == b.shape and np.array_equal(a, b)
        assert got[i][-1][1] is True and all(not c[1] for c in got[i][:-1])
