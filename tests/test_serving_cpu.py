"""CPU: the callers and formats either side of the hot path: WAV / PCM framing, the on-disk
voice-reference cache, the CLI (sub-commands, validation, the stdin serve loop) and the OpenAI-compatible endpoint with the
reference's request / response contract (examples/openai_server.py:77-265), over scripted model objects."""
import io
import json
import os
import struct
import wave

import numpy as np
import pytest
import torch

from fq3hip import audio_io, cli
from fq3hip.voice_cache import VoiceRefCache, cache_key, export_voice_clone_prompt


# ---- audio framing -----------------------------------------------------------------------------------------------
def test_wav_header_and_pcm16():
    h = audio_io.wav_header(24000)
    assert len(h) == 44 and h[:4] == b"RIFF" and h[8:12] == b"WAVE" and h[36:40] == b"data"
    assert struct.unpack("<I", h[4:8])[0] == 0xFFFFFFFF and struct.unpack("<I", h[40:44])[0] == 0xFFFFFFFF      # streaming: unknown size
    fmt = struct.unpack("<IHHIIHH", h[16:36])
    assert fmt == (16, 1, 1, 24000, 48000, 2, 16)
    pcm = np.array([0.0, 0.5, -0.5, 2.0, -2.0], np.float32)
    raw = np.frombuffer(audio_io.to_pcm16(pcm), dtype="<i2")
    assert raw.tolist() == [0, 16384, -16384, 32767, -32768]
    b = audio_io.to_wav_bytes(pcm, 24000)
    assert struct.unpack("<I", b[40:44])[0] == 10 and struct.unpack("<I", b[4:8])[0] == 46
    with wave.open(io.BytesIO(b)) as w:
        assert w.getframerate() == 24000 and w.getnframes() == 5


def test_write_read_wav_roundtrip(tmp_path):
    x = (np.sin(np.arange(2400) / 20.0) * 0.5).astype(np.float32)
    p = str(tmp_path / "sub" / "a.wav")
    audio_io.write_wav(p, x, 24000)
    y, sr = audio_io.read_wav(p)
    assert sr == 24000 and y.shape == x.shape and np.abs(y - x).max() < 1.0 / 32768 + 1e-6


def _extensible_wav(samples_bytes: bytes, channels: int, rate: int, bits: int, subformat_tag: int) -> bytes:
    """A WAVE_FORMAT_EXTENSIBLE file (40-byte fmt chunk with a SubFormat GUID), as ffmpeg / soundfile write 24- and 32-bit clips."""
    align = channels * bits // 8
    guid = struct.pack("<H", subformat_tag) + bytes.fromhex("000000001000800000aa00389b71")
    fmt = struct.pack("<HHIIHH", 0xFFFE, channels, rate, rate * align, align, bits) + struct.pack("<HHI", 22, bits, 3 if channels == 2 else 4) + guid
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 4) + b"abcd" + \
        b"data" + struct.pack("<I", len(samples_bytes)) + samples_bytes
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_read_wav_extensible_pcm_and_clear_errors(tmp_path):
    """WAVE_FORMAT_EXTENSIBLE with the PCM SubFormat (24-bit stereo, 16-bit mono) is read by the package's own RIFF parser (the
    standard library's `wave` refuses tag 0xFFFE); extensible IEEE float and unknown SubFormats fail with the clear message."""
    x = (np.sin(np.arange(480) / 9.0) * 0.4).astype(np.float64)
    v24 = np.round(x * 8388607).astype(np.int32)
    st = np.stack([v24, v24 // 2], axis=1).reshape(-1)                                   # stereo: right channel at half level
    raw24 = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in st)
    p = tmp_path / "e24.wav"
    p.write_bytes(_extensible_wav(raw24, 2, 24000, 24, 1))
    y, sr = audio_io.read_wav(str(p))
    assert sr == 24000 and y.shape == (480,) and np.abs(y - 0.75 * x).max() < 2e-6
    p16 = tmp_path / "e16.wav"
    p16.write_bytes(_extensible_wav(np.round(x * 32767).astype("<i2").tobytes(), 1, 16000, 16, 1))
    y, sr = audio_io.read_wav(str(p16))
    assert sr == 16000 and np.abs(y - x).max() < 1.0 / 32768 + 1e-6
    assert audio_io.load_audio(str(p16))[1] == 16000
    for tag, needle in ((3, "IEEE-float"), (0x55, "unknown SubFormat")):
        bad = tmp_path / f"bad{tag}.wav"
        bad.write_bytes(_extensible_wav(np.zeros(64, "<f4").tobytes(), 1, 24000, 32, tag))
        with pytest.raises(ValueError) as ei:
            audio_io.read_wav(str(bad))
        assert needle in str(ei.value) and "soundfile" in str(ei.value)


# ---- voice-reference cache -----------------------------------------------------------------------------------------
def test_voice_cache_roundtrip_and_staleness(tmp_path):
    audio = np.linspace(-1, 1, 24000, dtype=np.float32)
    key, meta = cache_key(audio, append_silence=True, model_identity="base-0b6")
    key2, _ = cache_key(audio, append_silence=False, model_identity="base-0b6")
    key3, _ = cache_key(audio * 0.5, append_silence=True, model_identity="base-0b6")
    assert len(key) == 64 and len({key, key2, key3}) == 3 and cache_key(audio, append_silence=True, model_identity="base-0b6")[0] == key
    c = VoiceRefCache(tmp_path / "voices")
    spk = torch.randn(1024)
    codes = torch.randint(0, 2048, (37, 16))
    assert c.load(key, meta) is None
    c.save(key, meta, spk, codes, ref_text="hello there")
    assert sorted(p.suffix for p in (tmp_path / "voices").iterdir()) == [".json", ".rvq", ".spk"]
    hit = c.load(key, meta)
    assert np.array_equal(hit["ref_spk_embedding"], spk.numpy()) and np.array_equal(hit["ref_code"], codes.numpy())
    assert hit["ref_text"] == "hello there"
    assert c.load(key, dict(meta, model_identity="other")) is None          # metadata mismatch = stale entry
    c.save(key2, cache_key(audio, append_silence=False, model_identity="base-0b6")[1], spk, None)       # x-vector only entry
    assert c.load(key2)["ref_code"] is None
    # upstream prompt item -> entry
    from types import SimpleNamespace
    k = export_voice_clone_prompt(c, audio, SimpleNamespace(ref_spk_embedding=spk, ref_code=codes, ref_text="t"), append_silence=True,
                                  model_identity="base-0b6")
    assert k == key


# ---- CLI ---------------------------------------------------------------------------------------------------------------
class _ScriptedModel:
    sample_rate = 24000

    def __init__(self):
        self.calls = []

    def _wave(self, text):
        return np.full(240 * max(1, len(text)), 0.25, np.float32)

    def generate_voice_clone(self, text, **kw):
        self.calls.append(("clone", text, kw))
        return [self._wave(text)], 24000

    def generate_voice_clone_streaming(self, text, chunk_size=12, **kw):
        self.calls.append(("clone_stream", text, kw))
        w = self._wave(text)
        for i in range(0, len(w), 1000):
            yield w[i:i + 1000], 24000, {"chunk_index": i // 1000}

    def generate_voice_clone_batch(self, texts, lanes=8, **kw):
        self.calls.append(("clone_batch", list(texts), dict(kw, lanes=lanes)))
        return [([self._wave(t)], 24000) for t in texts]

    def generate_custom_voice(self, text, speaker, **kw):
        self.calls.append(("custom", text, dict(kw, speaker=speaker)))
        return [self._wave(text)], 24000

    def generate_voice_design(self, text, instruct, **kw):
        self.calls.append(("design", text, dict(kw, instruct=instruct)))
        return [self._wave(text)], 24000


def test_cli_parser_and_validation(capsys):
    p = cli.build_parser()
    a = p.parse_args(["clone", "--text", "hi", "--output", "o.wav", "--ref-audio", "r.wav", "--ref-text", "t", "--streaming", "--chunk-size", "8"])
    assert a.mode == "clone" and a.streaming and a.chunk_size == 8 and a.temperature == 0.9 and a.top_k == 50 and a.repetition_penalty == 1.05
    for argv, msg in ((["clone", "--text", "x", "--output", "o.wav"], "requires --ref-audio"),
                      (["clone", "--text", "x", "--output", "o.wav", "--ref-audio", "r.wav"], "--ref-text is required"),
                      (["clone", "--text", "x", "--output", "o.wav", "--ref-spk", "a.spk"], "GGML backend"),
                      (["custom", "--text", "x", "--output", "o.wav"], "--speaker is required"),
                      (["design", "--text", "x", "--output", "o.wav"], "--instruct is required")):
        with pytest.raises(SystemExit) as e:
            cli.cmd_once(p.parse_args(argv), model=_ScriptedModel())
        assert e.value.code == 2 and msg in capsys.readouterr().out


def test_cli_once_and_serve_loop(tmp_path, capsys):
    p = cli.build_parser()
    m = _ScriptedModel()
    out = str(tmp_path / "one.wav")
    cli.cmd_once(p.parse_args(["clone", "--text", "hello", "--output", out, "--ref-audio", "r.wav", "--ref-text", "t"]), model=m)
    y, sr = audio_io.read_wav(out)
    assert sr == 24000 and len(y) == 240 * 5 and m.calls[0][0] == "clone" and m.calls[0][2]["do_sample"] is True
    cli.cmd_once(p.parse_args(["design", "--text", "abc", "--output", out, "--instruct", "calm", "--greedy"]), model=m)
    assert m.calls[-1][0] == "design" and m.calls[-1][2]["do_sample"] is False
    # serve: one line at a time (reference behaviour), streaming mode
    d = str(tmp_path / "serve1")
    a = p.parse_args(["serve", "--ref-audio", "r.wav", "--ref-text", "t", "--output-dir", d, "--streaming"])
    cli.cmd_serve(a, model=m, lines=["first\n", "\n", "second line\n", "quit\n", "never\n"])
    assert sorted(os.listdir(d)) == ["out_0001.wav", "out_0002.wav"]
    assert [c[0] for c in m.calls[-2:]] == ["clone_stream", "clone_stream"]
    # serve --lanes 3: waiting lines are decoded together through the batch entry point
    d2 = str(tmp_path / "serve2")
    a = p.parse_args(["serve", "--ref-audio", "r.wav", "--ref-text", "t", "--output-dir", d2, "--lanes", "3"])
    cli.cmd_serve(a, model=m, lines=["a\n", "bb\n", "ccc\n", "dddd\n", "exit\n"])
    assert sorted(os.listdir(d2)) == [f"out_000{i}.wav" for i in (1, 2, 3, 4)]
    batch_calls = [c for c in m.calls if c[0] == "clone_batch"]
    assert batch_calls[0][1] == ["a", "bb", "ccc"] and batch_calls[0][2]["lanes"] == 3
    assert "Wrote" in capsys.readouterr().out


# ---- OpenAI-compatible endpoint (lock scheduler, scripted model) -----------------------------------------------------------
def test_openai_speech_endpoint_contract():
    from fastapi.testclient import TestClient
    from fq3hip.server import create_app
    m = _ScriptedModel()
    voices = {"alloy": {"ref_audio": "a.wav", "ref_text": "t", "language": "English"}}
    client = TestClient(create_app(m, voices, default_voice="alloy", scheduler="lock"))
    assert client.get("/health").json()["status"] == "ok"
    r = client.post("/v1/audio/speech", json={"model": "tts-1", "input": "hello world", "voice": "alloy", "response_format": "wav"})
    assert r.status_code == 200 and r.headers["content-type"].startswith("audio/wav")
    body = r.content
    assert body[:4] == b"RIFF" and struct.unpack("<I", body[40:44])[0] == 0xFFFFFFFF        # streamed WAV: unknown length
    pcm = np.frombuffer(body[44:], dtype="<i2")
    assert len(pcm) == 240 * len("hello world") and abs(int(pcm[0]) - 8192) <= 1
    r = client.post("/v1/audio/speech", json={"input": "abc", "voice": "unknown-voice", "response_format": "pcm"})      # falls back to the default voice
    assert r.status_code == 200 and r.headers["content-type"].startswith("audio/pcm") and len(r.content) == 2 * 240 * 3
    assert m.calls[-1][2]["ref_audio"] == "a.wav" and m.calls[-1][2]["language"] == "English"
    assert client.post("/v1/audio/speech", json={"input": "   ", "voice": "alloy"}).status_code == 400
    assert client.post("/v1/audio/speech", json={"input": "x", "voice": "alloy", "response_format": "flac"}).status_code == 400
    client2 = TestClient(create_app(m, voices, default_voice=None, scheduler="lock"))
    assert client2.post("/v1/audio/speech", json={"input": "x", "voice": "nobody"}).status_code == 400


# ---- OpenAI-compatible endpoint, batch scheduler (scripted model + scripted lock-step decoder) ---------------------------------
class _ScriptedBatchModel(_ScriptedModel):
    """What BatchWorker needs from a model: _prepare_generation, _gen_kwargs, streaming_vocoder, _batch_decoder(...).run."""

    class _Voc:
        def __init__(self):
            self.n = 0

        def push(self, codes, ready_event=None):
            self.n += 1
            return np.full(100 * int(codes.shape[0]), 0.5, np.float32), 24000

    class _Dec:
        def __init__(self, owner):
            self.owner = owner

        def run(self, requests, on_error="raise", source=None, chunk_frames=None):
            import torch
            pending = list(requests)
            self.owner.calls.append(("run", chunk_frames))
            while pending:
                req = pending.pop(0)
                n = len(req.gen_kwargs["text"])
                if req.gen_kwargs["text"] == "boom":
                    yield req.rid, None, {"error": "RuntimeError('Input is too long')", "steps": 0}
                else:
                    done = 0
                    while n - done > chunk_frames:
                        done += chunk_frames
                        yield req.rid, torch.zeros(chunk_frames, 16, dtype=torch.long), {"is_final": False, "total_steps_so_far": done}
                    yield req.rid, torch.zeros(n - done, 16, dtype=torch.long), {"is_final": True, "total_steps_so_far": n, "steps": n}
                while source is not None:                    # requests that arrived meanwhile join the same run
                    r = source()
                    if r is None:
                        break
                    pending.append(r)

    def _prepare_generation(self, text, **kw):
        if text == "bad voice":
            raise ValueError("ref_text is required")
        self.calls.append(("prepare", text, kw))
        return None, None, None, text, None, None, None, None

    @staticmethod
    def _gen_kwargs(*a):
        return {}

    def streaming_vocoder(self, rc, chunk):
        return self._Voc()

    def _batch_decoder(self, lanes):
        return self._Dec(self)


def test_openai_endpoint_batch_scheduler_streams_and_reports_errors(monkeypatch):
    from fastapi.testclient import TestClient
    import fq3hip.batching as Bt
    from fq3hip.server import create_app

    class Req:                                            # BatchRequest stand-in keeping the text for the scripted decoder
        def __init__(self, rid, talker, tie, tam, tth, tpe, config, kw):
            self.rid, self.gen_kwargs = rid, dict(kw, text=tie)

    monkeypatch.setattr(Bt, "BatchRequest", Req)
    m = _ScriptedBatchModel()
    voices = {"alloy": {"voice_clone_prompt": {"x": 1}, "ref_text": "t", "language": "English", "chunk_size": 4}}
    client = TestClient(create_app(m, voices, default_voice="alloy", scheduler="batch", lanes=2, chunk_size=4))
    assert client.get("/health").json()["scheduler"] == "batch"
    r = client.post("/v1/audio/speech", json={"input": "ten chars!", "voice": "alloy", "response_format": "wav"})
    assert r.status_code == 200 and r.content[:4] == b"RIFF" and struct.unpack("<I", r.content[40:44])[0] == 0xFFFFFFFF
    pcm = np.frombuffer(r.content[44:], dtype="<i2")
    assert len(pcm) == 100 * 10 and abs(int(pcm[0]) - 16384) <= 1            # chunks of 4 + 4 + 2 frames, 100 samples each
    assert ("run", 4) in m.calls
    r = client.post("/v1/audio/speech", json={"input": "abcde", "voice": "alloy", "response_format": "pcm"})
    assert r.status_code == 200 and len(r.content) == 2 * 100 * 5
    # a request the decoder rejects, and one that fails before decoding, are 500s with the reason; the worker survives both
    r = client.post("/v1/audio/speech", json={"input": "boom", "voice": "alloy"})
    assert r.status_code == 500 and "too long" in r.text
    r = client.post("/v1/audio/speech", json={"input": "bad voice", "voice": "alloy"})
    assert r.status_code == 500 and "ref_text is required" in r.text
    r = client.post("/v1/audio/speech", json={"input": "ok", "voice": "alloy", "response_format": "pcm"})
    assert r.status_code == 200 and len(r.content) == 2 * 100 * 2


def test_voice_cache_key_is_one_construction_for_lookup_and_store(tmp_path):
    """A reference clip that is not at 24 kHz hits the entry its own analysis wrote (same loader + resampler on both sides), and an
    x-vector-only entry never serves an ICL request (the mode is part of the key)."""
    from types import SimpleNamespace
    from fq3hip.audio_io import write_wav
    from fq3hip.model import FasterQwen3TTS
    clip = tmp_path / "ref16k.wav"
    t = np.arange(16000, dtype=np.float32) / 16000.0
    write_wav(str(clip), 0.3 * np.sin(2 * np.pi * 220.0 * t), 16000)
    inner = SimpleNamespace(tts_model_type="base", tts_model_size="0b6", speech_tokenizer=None)
    m = FasterQwen3TTS(SimpleNamespace(model=inner, sample_rate=24000), None, None, device="cpu")
    m.set_voice_ref_cache(str(tmp_path / "voices"))
    spk, codes = torch.randn(1024), torch.randint(0, 2048, (21, 16))
    assert m._cached_voice_prompt(str(clip), "hi", False, True) is None
    # an x-vector-only analysis is written through ...
    m._store_voice_prompt(str(clip), SimpleNamespace(ref_spk_embedding=spk, ref_code=None, ref_text=None), True, True)
    hit = m._cached_voice_prompt(str(clip), "", True, True)
    assert hit is not None and hit[0]["x_vector_only_mode"] == [True] and torch.equal(hit[0]["ref_spk_embedding"][0], spk)
    # ... and does NOT answer the ICL request for the same clip (it used to, silently dropping ref_text)
    assert m._cached_voice_prompt(str(clip), "hi", False, False) is None
    assert m._cached_voice_prompt(str(clip), "hi", False, True) is None
    # the ICL analysis is then stored under its own key and found again: 16 kHz clip, resampled identically on both sides
    m._store_voice_prompt(str(clip), SimpleNamespace(ref_spk_embedding=spk, ref_code=codes, ref_text="hi"), False, True)
    vcp, rt = m._cached_voice_prompt(str(clip), "", False, True)
    assert vcp["icl_mode"] == [True] and torch.equal(vcp["ref_code"][0], codes) and rt == "hi"
    assert len(list((tmp_path / "voices").glob("*.json"))) == 2


def test_wav_loader_names_what_it_cannot_read(tmp_path):
    from fq3hip.audio_io import load_audio, write_wav
    good = tmp_path / "a.wav"
    write_wav(str(good), np.zeros(100, np.float32), 24000)
    a, sr = load_audio(str(good))
    assert sr == 24000 and a.shape == (100,)
    bad = tmp_path / "f.wav"           # IEEE-float WAV header (format tag 3)
    bad.write_bytes(b"RIFF" + struct.pack("<I", 36 + 16) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 1, 24000, 96000, 4, 32) +
                    b"data" + struct.pack("<I", 16) + np.zeros(4, "<f4").tobytes())
    try:
        import soundfile  # noqa: F401
    except ImportError:
        with pytest.raises(ValueError, match="IEEE-float"):
            load_audio(str(bad))


def test_batch_worker_answers_requests_the_scheduler_never_reported():
    """Whatever the decoder does, every submitted request gets an answer and DONE (no HTTP handler is left blocked)."""
    from fq3hip.server import BatchWorker

    class Model(_ScriptedBatchModel):
        class _Dec(_ScriptedBatchModel._Dec):
            def run(self, requests, on_error="raise", source=None, chunk_frames=None):
                for _r in requests:          # a scheduler that loses the request: no event at all
                    pass
                return
                yield

    w = BatchWorker(Model(), lanes=2, chunk_size=4)
    box = w.submit({"voice_clone_prompt": {"x": 1}, "ref_text": "t"}, "hello")
    first = box.get(timeout=10)
    assert isinstance(first, Exception) and "without an answer" in str(first)
    assert box.get(timeout=10) is BatchWorker.DONE
    w.inbox.put(None)
