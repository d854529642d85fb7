"""CPU: host-side decode loops (fq3hip/generate.py, streaming.py) against a scripted fake engine -- chunk
boundaries and flags (reference streaming.py:157-188), look-ahead issue order, noise-ring refills,
max_new_tokens / context clamps, timing-dict keys (generate.py:205-211)."""
from types import SimpleNamespace

import pytest
import torch

import fq3hip.generate as G
import fq3hip.streaming as S


class FakeEngine:
    """Mimics the Fq3Engine surface the loops use; 'generates' frame f as codes [f, f+1, ...] and stops
    (EOS) after `eos_after` frames."""

    def __init__(self, eos_after=10 ** 9, max_frames=64):
        self.dtype, self.device = torch.float32, torch.device("cpu")
        self.cfg = SimpleNamespace(num_code_groups=16, predictor=SimpleNamespace(vocab_size=32))
        self.max_frames = max_frames
        self.eos_after = eos_after
        self.issued = 0
        self.log = []

    def prefill(self, x, n_pad=0):
        self.log.append(("prefill", tuple(x.shape), n_pad))
        return torch.zeros(48), torch.zeros(x.shape[1])

    def sample(self, logits, **kw):
        self.log.append(("sample", kw["do_sample"], kw["suppress_eos"]))
        return torch.tensor([3])

    def decode_begin(self, **kw):
        self.begin = kw
        self.issued = 0

    def graph_capture(self):
        self.log.append(("capture",))

    def graph_reset(self):
        self.log.append(("reset",))

    def decode_frames(self, n):
        self.log.append(("frames", n))
        self.issued += n

    def decode_poll(self):
        n = min(self.issued, self.eos_after, self.begin["max_new_tokens"])
        self.log.append(("poll", n))
        return n, n >= self.eos_after

    def decode_codes(self, start, count):
        return torch.arange(start, start + count)[:, None] + torch.arange(16)[None, :]


class FakeTG:
    def __init__(self, eng):
        self.engine = eng
        self.state = None

    def prefill_kv(self, n):
        return n

    def set_generation_state(self, mask, deltas, n_pad=None):
        self.state = (None if mask is None else int((mask[0] == 0).sum()), deltas)


@pytest.fixture
def patched(monkeypatch):
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: SimpleNamespace(synchronize=lambda: None))
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: SimpleNamespace(record=lambda *a, **k: None))
    refills = []
    monkeypatch.setattr(G, "_refill", lambda eng, tn, pn: refills.append(eng.issued))
    return refills


def _args(eng, L=12, H=8):
    cfgT = SimpleNamespace(codec_eos_token_id=40, vocab_size=48)
    tie = torch.zeros(1, L, H); tam = torch.ones(1, L, dtype=torch.long)
    tth = torch.zeros(1, 3, H); tpe = torch.zeros(1, 1, H)
    pg = SimpleNamespace(do_sample=True)
    return (SimpleNamespace(rope_deltas=None), tie, tam, tth, tpe, cfgT, pg, FakeTG(eng))


def test_non_streaming_loop_polls_and_clamps(patched):
    eng = FakeEngine(eos_after=21)
    codes, timing = G.fast_generate(*_args(eng), max_new_tokens=100, min_new_tokens=2, poll_every=8)
    assert codes.shape == (21, 16) and set(timing) == {"prefill_ms", "decode_s", "steps", "ms_per_step", "steps_per_s"}
    assert timing["steps"] == 21
    assert eng.begin["max_new_tokens"] == 64                     # clamped to the context's frame capacity
    assert [e for e in eng.log if e[0] == "frames"] == [("frames", 8)] * 3          # stops polling once EOS is reported
    assert ("capture",) in eng.log and eng.begin["prefill_len"] == 12 and eng.begin["first_token"] == 3
    assert patched == [0]                                        # one ring fill (ring = 64 frames)
    eng2 = FakeEngine()
    codes, _ = G.fast_generate(*_args(eng2), max_new_tokens=5, parity_mode=True)
    assert codes.shape == (5, 16) and ("reset",) in eng2.log and ("capture",) not in eng2.log
    eng3 = FakeEngine(eos_after=0)
    codes, timing = G.fast_generate(*_args(eng3), max_new_tokens=5)
    assert codes is None and timing["steps"] == 0 and timing["ms_per_step"] == 0


def test_noise_ring_refills_on_ring_boundaries(patched):
    eng = FakeEngine(max_frames=200)
    G.fast_generate(*_args(eng), max_new_tokens=150, poll_every=40)
    assert patched == [0, 64, 128]
    # frames are never issued across a ring boundary in one call
    sizes = [e[1] for e in eng.log if e[0] == "frames"]
    assert sizes == [40, 24, 16, 40, 8, 22] and sum(sizes) == 150


def test_streaming_chunks_flags_and_lookahead(patched):
    eng = FakeEngine(eos_after=21)
    out = list(S.fast_generate_streaming(*_args(eng), max_new_tokens=100, chunk_size=8))
    metas = [(t["chunk_index"], t["chunk_steps"], t["total_steps_so_far"], t["is_final"]) for _, t in out]
    assert metas == [(0, 8, 8, False), (1, 8, 16, False), (2, 5, 21, True)]
    assert torch.equal(torch.cat([c for c, _ in out]), eng.decode_codes(0, 21))
    assert out[0][1]["prefill_ms"] >= 0 and out[1][1]["prefill_ms"] == 0
    # look-ahead: chunk k+1 is issued before chunk k is handed to the consumer, never after EOS
    order = [e[0] for e in eng.log if e[0] in ("frames", "poll")]
    assert order == ["frames", "poll", "frames", "poll", "frames", "poll"]
    # generation that ends exactly on a chunk boundary: no trailing partial chunk, last flag stays False
    eng2 = FakeEngine(eos_after=16)
    out2 = list(S.fast_generate_streaming(*_args(eng2), max_new_tokens=100, chunk_size=8))
    assert [(t["chunk_steps"], t["is_final"]) for _, t in out2] == [(8, False), (8, False)]
    # budget exhausted mid-chunk
    eng3 = FakeEngine()
    out3 = list(S.fast_generate_streaming(*_args(eng3), max_new_tokens=10, chunk_size=8))
    assert [(t["chunk_steps"], t["is_final"]) for _, t in out3] == [(8, False), (2, True)]
    # immediate EOS: nothing is yielded
    assert list(S.fast_generate_streaming(*_args(FakeEngine(eos_after=0)), max_new_tokens=10, chunk_size=8)) == []


def test_left_pad_and_greedy_paths(patched):
    eng = FakeEngine()
    args = list(_args(eng))
    args[2] = torch.tensor([[0, 0, 0] + [1] * 9])
    G.fast_generate(*args, max_new_tokens=4, do_sample=False, min_new_tokens=0)
    assert eng.log[0] == ("prefill", (12, 8), 3)
    assert ("sample", False, False) in eng.log
    assert eng.begin["talker_noise"] is None and eng.begin["pred_noise"] is not None      # predictor policy is separate
    assert args[7].state[0] == 3


def test_side_vocoder_synchronous_fallback_cuts_the_reference_share():
    """_SideVocoder with a foreign tokenizer (no decode_tensor): decodes through the upstream call and cuts the ICL reference's
    share of the waveform exactly like model.py:927-930; collect() returns the submissions in order."""
    import numpy as np
    import torch
    from fq3hip.model import _SideVocoder

    class Tok:
        sample_rate = 24000

        def decode(self, payload):
            codes = payload["audio_codes"][0]
            return [torch.arange(codes.shape[0] * 10, dtype=torch.float32)], 24000

    voc = _SideVocoder(Tok(), "cpu")
    assert not voc.async_ok
    voc.submit("a", torch.zeros(7, 16, dtype=torch.long))
    voc.submit("b", torch.zeros(10, 16, dtype=torch.long), ref_len=4)
    out = list(voc.collect())
    assert [k for k, _ in out] == ["a", "b"]
    assert len(out[0][1]) == 70 and out[0][1][0] == 0
    cut = int(4 / 10 * 100)
    assert len(out[1][1]) == 100 - cut and out[1][1][0] == cut
    assert list(voc.collect()) == []


def test_new_public_entry_points_keep_the_single_stream_vocabulary():
    import inspect
    from fq3hip.model import FasterQwen3TTS
    single = inspect.signature(FasterQwen3TTS.generate_voice_clone_streaming).parameters
    batch = inspect.signature(FasterQwen3TTS.generate_voice_clone_batch_streaming).parameters
    assert "language" in batch                      # one string or one per text; defaults to "English" like generate_voice_clone_batch
    for name in ("ref_audio", "ref_text", "max_new_tokens", "min_new_tokens", "temperature", "top_k", "top_p", "do_sample",
                 "repetition_penalty", "chunk_size", "xvec_only", "non_streaming_mode", "append_silence", "instruct", "voice_clone_prompt"):
        assert name in batch and batch[name].default == single[name].default, name
    assert list(batch)[1] == "texts" and batch["lanes"].default == 8
