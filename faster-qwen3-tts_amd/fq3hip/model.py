"""``FasterQwen3TTS``: the reference's public API (``faster_qwen3_tts/model.py``) over the MI355X HIP path.

Same constructor arguments, method names, argument order, defaults, return types and error
behaviour as the reference wrapper (pinned there by ``tests/test_voice_clone_prompt_api.py`` and
``tests/test_sample_rate.py``; mirrored here by ``tests/test_api_contract.py``), so callers switch by
changing the import.  What changed underneath: ``predictor_graph`` / ``talker_graph`` are HIP-backed
objects sharing one ``libfq3hip`` context, ``speech_tokenizer`` is the HIP codec decoder, and the
decode loop state lives on the GPU.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Any, Dict, Generator, List, Optional, Tuple, Union

import numpy as np
import torch

logger = logging.getLogger(__name__)

_GGML_ONLY = ("ref_spk", "ref_rvq", "ref_spk_emb", "ref_codes")


def _to_numpy(a) -> np.ndarray:
    if hasattr(a, "cpu"):
        return a.flatten().float().cpu().numpy()
    return a.flatten() if hasattr(a, "flatten") else a


class _SideVocoder:
    """Non-streaming vocoding of whole utterances off the decode stream (batched generation): ``submit`` enqueues the codec
    decode + the device-to-pinned-host copy on a side stream after the codes are ready, ``collect`` waits and returns the
    waveforms in submission order.  Tokenizers without ``decode_tensor`` (a foreign ``speech_tokenizer``) are decoded
    synchronously through the upstream call."""

    def __init__(self, tok, device, stream=None):
        self.tok = tok
        self.sample_rate = int(getattr(tok, "sample_rate", 24000))
        self.dev = torch.device(device) if not isinstance(device, torch.device) else device
        self.async_ok = torch.cuda.is_available() and hasattr(tok, "decode_tensor")
        if self.async_ok and stream is None:
            from .streams import concurrent_stream
            stream = concurrent_stream(self.dev)               # verified to run beside the decode stream (fq3hip/streams.py)
        self.stream = stream if self.async_ok else None
        self.items: list = []

    def _prefix(self, ref):
        """the tokenizer's cached front-end state of an ICL reference (built on the vocoder stream), or None"""
        if ref is None or not hasattr(self.tok, "prefix_for"):
            return None
        return self.tok.prefix_for(ref, self.stream)

    def submit(self, key, codes: torch.Tensor, ref_len: int = 0, ref=None) -> None:
        """``ref_len`` > 0: the first ``ref_len`` frames are the ICL reference; only the waveform after their share
        (``int(ref_len / T * n_samples)``, model.py:927-930) is produced.  ``ref``: that reference's own tensor (its cached
        front-end state then serves every utterance of the voice)."""
        if not self.async_ok or not hasattr(self.tok, "num_samples_total"):
            lst, _rate = self.tok.decode({"audio_codes": codes.unsqueeze(0)})
            a = _to_numpy(lst[0])
            self.items.append((key, a[int(ref_len / max(codes.shape[0], 1) * len(a)):] if ref_len > 0 else a, None, None))
            return
        cut = int(ref_len / max(codes.shape[0], 1) * self.tok.num_samples_total(codes.shape[0])) if ref_len > 0 else 0
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        codes.record_stream(self.stream)
        pf = self._prefix(ref) if ref_len > 0 else None
        with torch.cuda.stream(self.stream):
            pcm = self.tok.decode_tensor(codes, cut, prefix=pf) if pf is not None else self.tok.decode_tensor(codes, cut)
            host = torch.empty(pcm.shape, dtype=torch.float32, pin_memory=True)
            host.copy_(pcm, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.items.append((key, host, ev, pcm))          # pcm kept alive until its copy has completed

    MAX_GROUP = 16          # utterances per batched codec launch set (the workspace grows with it; the gain saturates well before)
    STREAM_GROUP = 32       # streaming CHUNKS per launch set (short inputs: 178-202 frames behind a reference, 33 in phase 2): measured per chunk
                            # 0.42 / 0.35 / 0.31 ms at 16 / 32 / 64 (profiles/r06_codec_time_bf16x2_groups.txt); 32 keeps the workspace at ~17 GB

    def add(self, key, codes: torch.Tensor, ref_len: int = 0, more: int = 0, ref=None) -> None:
        """Grouped form of ``submit``: utterances that finished in the same poll of the lock-step decode (``more`` = how many more of
        that poll follow, ``timing["more_in_poll"]``) are held back and vocoded TOGETHER -- utterances of equal length through one
        batched launch set (``decode_tensor_batch``: the [B, T, 16] form of the vocoder interface, model.py:924)."""
        self._held = getattr(self, "_held", [])
        self._held.append((key, codes, ref_len, ref))
        if more <= 0:
            held, self._held = self._held, []
            self.submit_many(held)

    def add_flush(self) -> None:
        held, self._held = getattr(self, "_held", []), []
        if held:
            self.submit_many(held)

    def submit_many(self, group) -> None:
        if not self.async_ok or not hasattr(self.tok, "decode_tensor_batch") or not hasattr(self.tok, "num_samples_total"):
            for key, codes, ref_len, ref in group:
                self.submit(key, codes, ref_len, ref)
            return
        classes: Dict[Any, list] = {}
        for key, codes, ref_len, ref in group:
            classes.setdefault((int(codes.shape[0]), int(ref_len)), []).append((key, codes, ref))
        for (T, ref_len), members in classes.items():
            for i in range(0, len(members), self.MAX_GROUP):
                part = members[i:i + self.MAX_GROUP]
                if len(part) == 1:
                    self.submit(part[0][0], part[0][1], ref_len, part[0][2])
                    continue
                cut = int(ref_len / max(T, 1) * self.tok.num_samples_total(T)) if ref_len > 0 else 0
                self.stream.wait_stream(torch.cuda.current_stream(self.dev))
                for _k, c, _r in part:
                    c.record_stream(self.stream)
                pfs = [self._prefix(r) for _k, _c, r in part] if ref_len > 0 else None
                if pfs is not None and any(p is None for p in pfs):
                    pfs = None
                with torch.cuda.stream(self.stream):
                    stacked = torch.stack([c for _k, c, _r in part])
                    pcm = self.tok.decode_tensor_batch(stacked, cut, prefixes=pfs) if pfs is not None else self.tok.decode_tensor_batch(stacked, cut)
                    host = torch.empty(pcm.shape, dtype=torch.float32, pin_memory=True)
                    host.copy_(pcm, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                for j, (key, _c, _r) in enumerate(part):
                    self.items.append((key, host[j], ev, pcm))

    # ---- incremental (exact) vocoding of utterances that are still decoding ------------------------------------------------
    # The codec decoder is causal: decode(codes[:p]) is bit for bit the prefix of decode(codes), and a tail decode from sample s is
    # the tail of the full decode.  So the waveform of a running utterance can be produced in slices -- every ``inc`` frames the
    # samples that became final, by ``decode_tensor_batch(codes so far, first_sample = samples already produced)`` -- on the side
    # stream, under the lock-step decode of the following frames, and across all lanes that reached the boundary together as ONE
    # batched launch set.  What is left when the utterance ends is the last slice, not the whole waveform.  The concatenation
    # is exactly ``decode_tensor(ref + codes)[cut:]``; the reference's cut of an ICL waveform (``int(ref_len / T * n_samples)``,
    # model.py:927-930) depends on the final length T, so slices start at a lower bound of it and the few samples in front of
    # the real cut are dropped at the end.
    def _cut_floor(self, ref_len: int, max_total: int) -> int:
        if ref_len <= 0:
            return 0
        key = (ref_len, max_total)
        cache = self.__dict__.setdefault("_cut_cache", {})
        if key not in cache:
            cache[key] = min(int(ref_len / T * self.tok.num_samples_total(T)) for T in range(ref_len + 1, max(ref_len + 2, max_total + 1)))
        return cache[key]

    def inc_add(self, key, chunk, ref_codes, ready_event, final: bool, more: int, max_new: int) -> None:
        """``chunk``: the frames of utterance ``key`` that completed since the last call (LongTensor[n, 16] on the device, or None / empty).
        ``final``: the utterance is over.  ``more``: events of the same poll still to come (the slice is decoded when it reaches 0)."""
        inc = self.__dict__.setdefault("_inc", {})
        st = inc.get(key)
        if st is None:
            ref_len = int(ref_codes.shape[0]) if ref_codes is not None else 0
            st = inc[key] = dict(ref=ref_codes, ref_len=ref_len, chunks=[], T=0, done=0, parts=[], final=False,
                                 floor=self._cut_floor(ref_len, ref_len + int(max_new)))
        if chunk is not None and chunk.shape[0] > 0:
            chunk.record_stream(self.stream)
            st["chunks"].append(chunk)
            st["T"] += int(chunk.shape[0])
            self.__dict__.setdefault("_inc_held", []).append((key, ready_event))
        st["final"] = st["final"] or bool(final)
        if more <= 0:
            self.inc_flush()

    def inc_flush(self) -> None:
        held, self._inc_held = self.__dict__.get("_inc_held", []), []
        if not held:
            return
        main = torch.cuda.current_stream(self.dev)
        jobs, seen = {}, set()
        for key, ev in held:
            if ev is not None:
                self.stream.wait_event(ev)
            else:
                self.stream.wait_stream(main)
            if key in seen:
                continue
            seen.add(key)
            st = self._inc[key]
            T = st["ref_len"] + st["T"]
            first = st["floor"] + st["done"]
            n = self.tok.num_samples_total(T)
            if first < n:
                jobs.setdefault((T, first), []).append((key, st, n))
        with torch.cuda.stream(self.stream):
            for (T, first), members in jobs.items():
                for i in range(0, len(members), self.MAX_GROUP):
                    part = members[i:i + self.MAX_GROUP]
                    fulls = [torch.cat(([st["ref"].to(st["chunks"][0].device)] if st["ref"] is not None else []) + st["chunks"], dim=0)
                             for _k, st, _n in part]
                    if len(part) == 1:
                        pcm = self.tok.decode_tensor(fulls[0], first).unsqueeze(0)
                    else:
                        pcm = self.tok.decode_tensor_batch(torch.stack(fulls), first)
                    host = torch.empty(pcm.shape, dtype=torch.float32, pin_memory=True)
                    host.copy_(pcm, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                    for j, (_k, st, n) in enumerate(part):
                        st["parts"].append((host[j], ev, pcm))
                        st["done"] = n - st["floor"]

    def inc_collect(self):
        """-> (key, waveform) of every utterance handed to ``inc_add``, in first-seen order (waits for the side stream)."""
        self.inc_flush()
        inc, self._inc = self.__dict__.get("_inc", {}), {}
        for key, st in inc.items():
            if not st["parts"]:
                yield key, np.zeros(1, dtype=np.float32)
                continue
            rows = []
            for host, ev, _pcm in st["parts"]:
                ev.synchronize()
                rows.append(host.numpy())
            a = np.concatenate(rows) if len(rows) > 1 else rows[0].copy()
            T = st["ref_len"] + st["T"]
            cut = int(st["ref_len"] / max(T, 1) * self.tok.num_samples_total(T)) if st["ref_len"] > 0 else 0
            yield key, a[cut - st["floor"]:]

    def collect(self):
        for key, host, ev, _pcm in self.items:
            if ev is None:
                yield key, host
            else:
                ev.synchronize()
                yield key, host.numpy().copy()
        self.items = []


class StreamingVocoder:
    """The reference's streaming windowing (``faster_qwen3_tts/model.py:1048-1137``) as a state machine over code chunks:
    phase 1 re-decodes everything (reference codes in front) until >= max(25, chunk) frames exist and calibrates samples
    per frame; phase 2 decodes [25 context frames + the new chunk] and drops the context.  With the HIP tokenizer only the
    samples that are kept are produced (``decode_tensor(codes, first_sample)``: bit-identical to slicing the full decode)
    and the codec runs on ``side_stream`` so that it overlaps the next frames' decode kernels."""

    CONTEXT_FRAMES = 25

    def __init__(self, tok, ref_codes, chunk_size: int, device, side_stream=None):
        self.tok, self.ref_codes, self.side = tok, ref_codes, side_stream
        self.dev = torch.device(device) if not isinstance(device, torch.device) else device
        self.min_cal = max(self.CONTEXT_FRAMES, int(chunk_size))
        self.all_codes: List[torch.Tensor] = []
        self.prev_len, self.spf = 0, None
        # phase 1 decodes ``ref_codes + everything so far`` for every chunk: with the tokenizer's cached front-end state of the
        # reference (built once per voice, on the vocoder stream) only the rows behind it are computed -- bit-identical
        self.prefix = (tok.prefix_for(ref_codes, side_stream) if side_stream is not None and ref_codes is not None and hasattr(tok, "prefix_for")
                       else None)
        self.last_prefix = None          # the prefix the decode returned by the last ``prepare`` may use (None in phase 2)

    def _vocode(self, codes_in, first_sample, ev, prefix=None):
        """waveform[first_sample:] of codes_in (host array)."""
        if self.side is None:
            lst, rate = self.tok.decode({"audio_codes": codes_in.unsqueeze(0)})
            return _to_numpy(lst[0])[first_sample:], rate
        if ev is not None:
            self.side.wait_event(ev)
        else:
            self.side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.side):
            # (.cpu() synchronises the side stream only)
            out = _to_numpy(self.tok.decode_tensor(codes_in, first_sample, prefix=prefix) if prefix is not None else self.tok.decode_tensor(codes_in, first_sample))
        return out, self.tok.sample_rate

    def _ref_on(self, device):
        """The reference codes on ``device``.  A host tensor is uploaded ONCE per tensor object and device (noted on the object: the
        streams of one voice share the prompt's tensor), not once per chunk of every stream (a synchronous copy each)."""
        rc = self.ref_codes
        if rc.device == device:
            return rc
        note = getattr(rc, "fq3_on_device", None)
        if not (isinstance(note, tuple) and note[0] == device and (note[2] is None or (not rc.is_inference() and note[2] == rc._version))):
            note = (device, rc.to(device), None if rc.is_inference() else rc._version)
            try:
                rc.fq3_on_device = note
            except Exception:
                pass
        return note[1]

    def _cat(self, parts, ev):
        if self.side is None:
            return torch.cat(parts, dim=0)
        with torch.cuda.stream(self.side):
            if ev is not None:
                self.side.wait_event(ev)
            else:
                self.side.wait_stream(torch.cuda.current_stream(self.dev))
            return torch.cat(parts, dim=0)

    def prepare(self, chunk: torch.Tensor, ready_event=None):
        """Side-stream form of ``push`` in two halves: advances the windowing state by ``chunk`` and returns ``(codes_in,
        first_sample)`` -- the decode whose ``waveform[first_sample:]`` is this chunk's audio -- so that a caller with several
        utterances at the same point (the first chunks of lock-step lanes) can run those decodes as ONE batched launch set."""
        assert self.side is not None
        chunk.record_stream(self.side)
        self.all_codes.append(chunk)
        n_new = chunk.shape[0]
        flat = self._cat(self.all_codes, ready_event)
        n_total = flat.shape[0]
        if self.spf is None:
            rc = self.ref_codes
            inp = self._cat([self._ref_on(flat.device), flat], ready_event) if rc is not None else flat
            ref_len = rc.shape[0] if rc is not None else 0
            n_audio = self.tok.num_samples_total(inp.shape[0])
            cut = int(ref_len / max(inp.shape[0], 1) * n_audio) if ref_len else 0
            first = cut + self.prev_len
            self.prev_len = n_audio - cut
            if n_total >= self.min_cal:
                self.spf = self.prev_len / n_total
            self.last_prefix = self.prefix if ref_len else None
            return inp, first
        start = max(0, n_total - n_new - self.CONTEXT_FRAMES)
        window = flat[start:]
        n_ctx = window.shape[0] - n_new
        self.last_prefix = None
        return window, (int(round(n_ctx * self.spf)) if n_ctx > 0 else 0)

    def push(self, chunk: torch.Tensor, ready_event=None):
        """``chunk`` LongTensor[n_new, 16] (device) -> (new audio as a host array, sample_rate)."""
        if self.side is not None:
            inp, first = self.prepare(chunk, ready_event)
            return self._vocode(inp, first, ready_event, self.last_prefix)
        self.all_codes.append(chunk)
        n_new = chunk.shape[0]
        flat = self._cat(self.all_codes, ready_event)
        n_total = flat.shape[0]
        if self.spf is None:
            rc = self.ref_codes
            inp = self._cat([self._ref_on(flat.device), flat], ready_event) if rc is not None else flat
            ref_len = rc.shape[0] if rc is not None else 0
            if self.side is not None:
                # model.py:1095-1100 without materialising what is thrown away: audio[cut:][prev_len:]
                n_audio = self.tok.num_samples_total(inp.shape[0])
                cut = int(ref_len / max(inp.shape[0], 1) * n_audio) if ref_len else 0
                new_audio, sr = self._vocode(inp, cut + self.prev_len, ready_event)
                gen_len = n_audio - cut
            else:
                audio, sr = self._vocode(inp, 0, ready_event)
                if ref_len:
                    audio = audio[int(ref_len / max(inp.shape[0], 1) * len(audio)):]
                new_audio = audio[self.prev_len:]
                gen_len = len(audio)
            self.prev_len = gen_len
            if n_total >= self.min_cal:
                self.spf = gen_len / n_total
        else:
            start = max(0, n_total - n_new - self.CONTEXT_FRAMES)
            window = flat[start:]
            n_ctx = window.shape[0] - n_new
            new_audio, sr = self._vocode(window, int(round(n_ctx * self.spf)) if n_ctx > 0 else 0, ready_event)
        return new_audio, sr


class FasterQwen3TTS:
    """Qwen3-TTS with a hipGraph-captured, hand-written HIP decode path (drop-in for the CUDA-graph wrapper)."""

    def __init__(self, base_model, predictor_graph, talker_graph, device: str = "cuda",
                 dtype: torch.dtype = torch.bfloat16, max_seq_len: int = 2048):
        self.model = base_model
        self.predictor_graph = predictor_graph
        self.talker_graph = talker_graph
        self.device = device
        self.dtype = dtype
        self.max_seq_len = max_seq_len
        self.sample_rate = self._infer_sample_rate(base_model)
        self._warmed_up = False
        self._voice_prompt_cache: Dict[Any, Any] = {}

    # ---- small helpers pinned by the reference's unit tests ------------------------------------------------
    @staticmethod
    def _get_speech_tokenizer(base_model):
        return getattr(getattr(base_model, "model", None), "speech_tokenizer", None)

    @property
    def speech_tokenizer(self):
        tok = self._get_speech_tokenizer(self.model)
        if tok is None:
            raise AttributeError("Underlying model does not expose a speech_tokenizer")
        return tok

    @staticmethod
    def _infer_sample_rate(base_model) -> int:
        """speech_tokenizer.sample_rate, then base_model.sample_rate, then 24000 (model.py:59-84)."""
        rate = None
        tok = FasterQwen3TTS._get_speech_tokenizer(base_model)
        if tok is not None:
            rate = getattr(tok, "sample_rate", None)
        if rate is None:
            rate = getattr(base_model, "sample_rate", None)
        if rate is None:
            logger.warning("Could not infer sample rate from base model; defaulting to 24000 Hz.")
            return 24000
        return int(rate)

    @staticmethod
    def _resolve_non_streaming_mode(non_streaming_mode: Optional[bool], *, default: bool) -> bool:
        return default if non_streaming_mode is None else non_streaming_mode

    @staticmethod
    def _reject_ggml_cached_reference_args(ref_spk=None, ref_rvq=None, ref_spk_emb=None, ref_codes=None) -> None:
        if any(v is not None for v in (ref_spk, ref_rvq, ref_spk_emb, ref_codes)):
            raise NotImplementedError(
                "ref_spk/ref_rvq cached qwentts.cpp references require backend='ggml'. "
                "Use voice_clone_prompt for precomputed prompts with the torch backend.")

    # ---- construction ---------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, model_name: str, device: str = "cuda", dtype: Union[str, torch.dtype] = torch.bfloat16,
                        attn_implementation: str = "sdpa", max_seq_len: int = 2048, backend: str = "torch",
                        quant: str = "BF16", gguf_talker_path: Optional[Union[str, Path]] = None,
                        gguf_codec_path: Optional[Union[str, Path]] = None,
                        qwentts_library_path: Optional[Union[str, Path]] = None, qwentts_use_fa: bool = True,
                        qwentts_clamp_fp16: bool = False, qwentts_ref_cache_dir: Optional[Union[str, Path]] = None,
                        cache_dir: Optional[Union[str, Path]] = None, local_files_only: bool = False,
                        codec_precision: Optional[str] = None):
        """Load a local Qwen3-TTS checkpoint directory and build the HIP decode context.

        ``backend`` keeps the reference's vocabulary: ``"torch"`` (the default) selects the graph-captured
        fast path -- here the HIP one; ``"ggml"``/``"qwentts"`` name the external qwentts.cpp runtime, which
        has no ROCm build in this repository.  ``attn_implementation`` is accepted for compatibility (the
        HIP attention kernel is the only implementation)."""
        if backend not in ("torch", "ggml", "qwentts"):
            raise ValueError(f"Unsupported backend {backend!r}. Expected 'torch', 'ggml', or 'qwentts'.")
        if backend in ("ggml", "qwentts"):
            raise NotImplementedError("the qwentts.cpp (GGML) backend is not part of the MI355X build; use backend='torch'")
        if isinstance(dtype, str):
            dtype = getattr(torch, dtype)
        if not device.startswith("cuda") or not torch.cuda.is_available():
            raise ValueError("CUDA graphs require CUDA device")       # same message as the reference (model.py:181-182)
        from .weights import load_hf_checkpoint
        import os
        path = str(model_name)
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"{model_name!r} is not a local checkpoint directory; this build has no network access "
                "(HF_HUB_OFFLINE) -- download the model and pass its path.")
        cfg, weights = load_hf_checkpoint(path, dtype=dtype, device="cpu")
        tokenizer = None
        try:
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(path, local_files_only=True)
        except Exception as e:      # tokenizer files are optional for code-only workflows
            logger.warning("no text tokenizer loaded from %s (%s)", path, e)
        return cls.from_weights(cfg, weights, device=device, dtype=dtype, max_seq_len=max_seq_len, tokenizer=tokenizer,
                                codec_precision=codec_precision)

    @classmethod
    def from_weights(cls, cfg, weights, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                     max_seq_len: int = 2048, tokenizer=None, codec_max_frames: int = 1024, max_frames: int = 2048,
                     share: Optional["FasterQwen3TTS"] = None, codec_precision: Optional[str] = None):
        """Build from an in-memory weight table (real or seeded synthetic, ``fq3hip.weights``).
        ``share``: an existing model on the same GPU whose weight replica this instance borrows -- one
        instance (decode context + codec workspace) per concurrently running utterance.
        ``codec_precision``: ``None`` = model dtype, ``"fp32"`` = high-precision vocoder (see ``NativeQwen3TTS``)."""
        if not str(device).startswith("cuda") or not torch.cuda.is_available():
            raise ValueError("CUDA graphs require CUDA device")
        from .native_model import NativeQwen3TTS
        from .predictor_graph import PredictorGraph
        from .talker_graph import TalkerGraph
        base = NativeQwen3TTS(cfg, weights, device=device, dtype=dtype, max_seq_len=max_seq_len, tokenizer=tokenizer,
                              codec_max_frames=codec_max_frames, max_frames=max_frames,
                              share=share.model if share is not None else None, codec_precision=codec_precision)
        pg = PredictorGraph(base.engine, do_sample=True, top_k=50, temperature=0.9)      # model.py:209-218
        tg = TalkerGraph(base.engine)
        return cls(base_model=base, predictor_graph=pg, talker_graph=tg, device=device, dtype=dtype,
                   max_seq_len=max_seq_len)

    def warmup(self, prefill_len: int = 100) -> None:
        """Capture the decode hipGraph once (idempotent, model.py:239-252)."""
        if self._warmed_up:
            return
        self.predictor_graph.capture(num_warmup=3)
        self.talker_graph.capture(prefill_len=prefill_len, num_warmup=3)
        self._warmed_up = True

    def _warmup(self, prefill_len: int) -> None:
        self.warmup(prefill_len=prefill_len)

    def generate(self, text: str, language: str = "English", max_new_tokens: int = 2048, temperature: float = 0.9,
                 top_k: int = 50, do_sample: bool = True, repetition_penalty: float = 1.05) -> Tuple[list, int]:
        raise NotImplementedError("Default voice generation not yet implemented. "
                                  "Use generate_voice_clone() with reference audio.")

    # ---- voice-clone prompt resolution (model.py:295-463) -------------------------------------------------------
    def _resolve_voice_clone_prompt(self, input_ids, ref_audio, ref_text: str, xvec_only: bool, append_silence: bool,
                                    voice_clone_prompt):
        if voice_clone_prompt is not None:
            return self._resolve_precomputed_voice_clone_prompt(input_ids=input_ids, ref_text=ref_text,
                                                                voice_clone_prompt=voice_clone_prompt)
        if ref_audio is None:
            raise ValueError("ref_audio is required when voice_clone_prompt is not provided")
        return self._resolve_voice_clone_prompt_from_reference(input_ids=input_ids, ref_audio=ref_audio,
                                                               ref_text=ref_text, xvec_only=xvec_only,
                                                               append_silence=append_silence)

    def _ref_ids(self, text: str):
        return self.model._tokenize_texts([self.model._build_ref_text(text)])[0]

    def _resolve_precomputed_voice_clone_prompt(self, input_ids, ref_text: str, voice_clone_prompt):
        n = len(input_ids)
        if isinstance(voice_clone_prompt, list):
            if len(voice_clone_prompt) != n:
                raise ValueError(f"voice_clone_prompt must have length {n}, got {len(voice_clone_prompt)}")
            vcp = self.model._prompt_items_to_voice_clone_prompt(voice_clone_prompt)
            ref_ids = []
            for item in voice_clone_prompt:
                if bool(item.icl_mode):
                    text = item.ref_text if item.ref_text else ref_text
                    if not text:
                        raise ValueError("ref_text is required when voice_clone_prompt uses ICL mode.")
                    ref_ids.append(self._ref_ids(text))
                else:
                    ref_ids.append(None)
            return vcp, ref_ids, any(vcp["icl_mode"])
        missing = [k for k in ("ref_spk_embedding",) if k not in voice_clone_prompt]
        if missing:
            raise ValueError(f"voice_clone_prompt missing required keys: {missing}. Expected keys: ['ref_spk_embedding']")
        for key in ("ref_spk_embedding", "x_vector_only_mode", "icl_mode", "ref_code"):
            if key in voice_clone_prompt:
                v = voice_clone_prompt[key]
                if not isinstance(v, list) or len(v) != n:
                    raise ValueError(f"voice_clone_prompt[{key!r}] must be a list with length {n}")
        xvec = voice_clone_prompt.get("x_vector_only_mode", [True] * n)
        if "icl_mode" in voice_clone_prompt:
            icl = [bool(v) for v in voice_clone_prompt["icl_mode"]]
            for i, (a, b) in enumerate(zip(xvec, icl)):
                if bool(a) == bool(b):
                    raise ValueError(f"voice_clone_prompt has inconsistent mode flags at index {i}: "
                                     "x_vector_only_mode and icl_mode must be opposites")
        else:
            icl = [not bool(v) for v in xvec]
        codes = voice_clone_prompt.get("ref_code", [None] * n)
        for i, (a, b, c) in enumerate(zip(xvec, icl, codes)):
            if bool(a) and c is not None:
                raise ValueError(f"voice_clone_prompt index {i}: ref_code must be None in x_vector_only mode")
            if bool(b) and c is None:
                raise ValueError(f"voice_clone_prompt index {i}: ref_code is required in ICL mode")
        vcp = dict(ref_code=codes, ref_spk_embedding=voice_clone_prompt["ref_spk_embedding"],
                   x_vector_only_mode=[bool(v) for v in xvec], icl_mode=[bool(v) for v in icl])
        using_icl = any(vcp["icl_mode"])
        if using_icl:
            if not ref_text:
                raise ValueError("ref_text is required when voice_clone_prompt uses ICL mode.")
            rid = self._ref_ids(ref_text)
            ref_ids = [rid if f else None for f in vcp["icl_mode"]]
        else:
            ref_ids = [None] * n
        return vcp, ref_ids, using_icl

    def _load_ref_audio_with_silence(self, ref_audio, silence_secs: float = 0.5):
        """model.py:278-293.  ``soundfile`` when it is installed, otherwise the standard-library WAV reader."""
        from .audio_io import load_audio
        audio, sr = load_audio(ref_audio)
        if silence_secs > 0:
            audio = np.concatenate([audio, np.zeros(int(silence_secs * sr), dtype=np.float32)])
        return audio, sr

    def set_voice_ref_cache(self, directory) -> None:
        """Serve ``ref_audio=...`` calls from an on-disk cache of precomputed voice-clone prompts (``fq3hip/voice_cache.py``;
        the role the reference's ``qwentts_ref_cache_dir`` plays for its GGML backend, ``ggml_backend.py:403-471``)."""
        from .voice_cache import VoiceRefCache
        self._voice_ref_cache = VoiceRefCache(directory) if directory else None

    def _voice_cache_entry(self, ref_audio, xvec_only: bool, append_silence: bool):
        """(audio at 24 kHz, key, metadata) of a reference clip: ONE construction for lookups and stores (same loader, same
        resampler, the request's mode in the metadata), so that what is written is what is found."""
        from .audio_io import resample
        from .voice_cache import cache_key
        silence = 0.5 if (append_silence and not xvec_only) else 0.0
        audio, sr = self._load_ref_audio_with_silence(ref_audio, silence_secs=silence)
        audio = resample(audio, sr, 24000)
        ident = f"{getattr(self.model.model, 'tts_model_type', 'base')}-{getattr(self.model.model, 'tts_model_size', '')}"
        key, meta = cache_key(audio, append_silence=silence > 0, model_identity=ident, mode="xvec" if xvec_only else "icl")
        return audio, key, meta, silence > 0, ident

    def _cached_voice_prompt(self, ref_audio, ref_text: str, xvec_only: bool, append_silence: bool):
        cache = getattr(self, "_voice_ref_cache", None)
        if cache is None:
            return None
        _audio, key, meta, _sil, _ident = self._voice_cache_entry(ref_audio, xvec_only, append_silence)
        hit = cache.load(key, meta)
        if hit is None or (not xvec_only and hit["ref_code"] is None):      # an entry without codes cannot serve an ICL request
            return None
        spk = torch.from_numpy(hit["ref_spk_embedding"])
        if xvec_only:
            return dict(ref_code=[None], ref_spk_embedding=[spk], x_vector_only_mode=[True], icl_mode=[False]), None
        return (dict(ref_code=[torch.from_numpy(hit["ref_code"])], ref_spk_embedding=[spk], x_vector_only_mode=[False],
                     icl_mode=[True]), hit["ref_text"] or ref_text)

    def _store_voice_prompt(self, ref_audio, item, xvec_only: bool, append_silence: bool) -> None:
        """Write a freshly analysed reference through to the on-disk cache (when one is set), so that the next process
        serves it without running the analysers (the reference's GGML backend does the same, ggml_backend.py:448-465)."""
        cache = getattr(self, "_voice_ref_cache", None)
        if cache is None:
            return
        from .voice_cache import export_voice_clone_prompt
        audio, _key, _meta, sil, ident = self._voice_cache_entry(ref_audio, xvec_only, append_silence)
        export_voice_clone_prompt(cache, audio, item, append_silence=sil, model_identity=ident,
                                  ref_text=getattr(item, "ref_text", None) or "")

    def _resolve_voice_clone_prompt_from_reference(self, input_ids, ref_audio, ref_text: str, xvec_only: bool,
                                                   append_silence: bool):
        using_icl = not xvec_only
        key = (str(ref_audio), ref_text, xvec_only, append_silence)
        if key in self._voice_prompt_cache:
            vcp, ref_ids = self._voice_prompt_cache[key]
            return vcp, ref_ids, using_icl
        cached = self._cached_voice_prompt(ref_audio, ref_text, xvec_only, append_silence)
        if cached is not None:
            vcp, rt = cached
            using_icl = bool(vcp["icl_mode"][0])
            if using_icl and not rt:
                raise ValueError("ref_text is required when voice_clone_prompt uses ICL mode.")
            ref_ids = [self._ref_ids(rt) if using_icl else None]
        elif xvec_only:
            items = self.model.create_voice_clone_prompt(ref_audio=str(ref_audio), ref_text="", x_vector_only_mode=True)
            vcp = dict(ref_code=[None], ref_spk_embedding=[items[0].ref_spk_embedding], x_vector_only_mode=[True],
                       icl_mode=[False])
            ref_ids = [None] * len(input_ids)
            self._store_voice_prompt(ref_audio, items[0], xvec_only, append_silence)
        else:
            audio = self._load_ref_audio_with_silence(ref_audio, silence_secs=0.5 if append_silence else 0.0)
            items = self.model.create_voice_clone_prompt(ref_audio=audio, ref_text=ref_text)
            vcp = self.model._prompt_items_to_voice_clone_prompt(items)
            rt = items[0].ref_text
            ref_ids = [self._ref_ids(rt) if rt else None]
            self._store_voice_prompt(ref_audio, items[0], xvec_only, append_silence)
        self._voice_prompt_cache[key] = (vcp, ref_ids)
        return vcp, ref_ids, using_icl

    # ---- prompt assembly (host glue; layout = SURVEY.md Appendix B / model.py:583-805) --------------------------
    def _build_talker_inputs_local(self, m, input_ids, ref_ids, voice_clone_prompt, languages, speakers,
                                   non_streaming_mode: bool, instruct_ids=None):
        tk, tc, mc = m.talker, m.config.talker_config, m.config
        if len(input_ids) == 1 and getattr(tk, "hip_prompt_ready", False):
            # native model: the whole prompt is three launches of HIP arithmetic (fq3hip/prompt.py); the tensor-op
            # version below is kept for foreign `m` objects (an upstream qwen-tts model) and for batches
            from .prompt import build_talker_inputs_hip
            vcp = voice_clone_prompt
            return build_talker_inputs_hip(m, input_ids[0], ref_ids[0] if ref_ids else None, vcp, 0, languages[0],
                                           speakers[0] if speakers is not None else None, non_streaming_mode,
                                           instruct_ids[0] if instruct_ids is not None else None)
        dev = tk.device
        emb_c = tk.get_input_embeddings()
        text = lambda ids: tk.text_projection(tk.get_text_embeddings()(ids))
        ids_t = lambda rows: torch.tensor(rows, device=dev, dtype=torch.long)
        spk_embeds = m.generate_speaker_prompt(voice_clone_prompt) if voice_clone_prompt is not None else None
        speakers = speakers if speakers is not None else [None] * len(input_ids)
        seqs, trailing, pad = [], [], None
        for i, (iid, lang, spk) in enumerate(zip(input_ids, languages, speakers)):
            parts = []
            if instruct_ids is not None and instruct_ids[i] is not None:
                parts.append(text(instruct_ids[i]))
            if spk_embeds is not None:
                use = voice_clone_prompt["x_vector_only_mode"][i] or voice_clone_prompt["icl_mode"][i]
                spk_e = spk_embeds[i] if use else None
            elif spk in ("", None):
                spk_e = None
            else:
                if spk.lower() not in tc.spk_id:
                    raise NotImplementedError(f"Speaker {spk} not implemented")
                spk_e = emb_c(ids_t(tc.spk_id[spk.lower()]))
            assert lang is not None
            if lang.lower() == "auto":
                lang_id = None
            else:
                if lang.lower() not in tc.codec_language_id:
                    raise NotImplementedError(f"Language {lang} not implemented")
                lang_id = tc.codec_language_id[lang.lower()]
            if lang.lower() in ("chinese", "auto") and spk not in ("", None) and tc.spk_is_dialect.get(spk.lower()):
                lang_id = tc.codec_language_id[tc.spk_is_dialect[spk.lower()]]
            bos, eos, pad = text(ids_t([[mc.tts_bos_token_id, mc.tts_eos_token_id, mc.tts_pad_token_id]])).chunk(3, dim=1)
            prefix = ([tc.codec_nothink_id, tc.codec_think_bos_id, tc.codec_think_eos_id] if lang_id is None else
                      [tc.codec_think_id, tc.codec_think_bos_id, lang_id, tc.codec_think_eos_id])
            cod = [emb_c(ids_t([prefix]))]
            if spk_e is not None:
                cod.append(spk_e.view(1, 1, -1))
            cod.append(emb_c(ids_t([[tc.codec_pad_id, tc.codec_bos_id]])))
            cod = torch.cat(cod, dim=1)                                   # [..., codec_pad, codec_bos]
            role = text(iid[:, :3])
            head = torch.cat((pad.expand(-1, cod.shape[1] - 2, -1), bos), dim=1) + cod[:, :-1]
            parts += [role, head]
            icl = (voice_clone_prompt is not None and voice_clone_prompt.get("ref_code", None) is not None
                   and voice_clone_prompt["icl_mode"][i])
            if icl:
                icl_embed, trail = m.generate_icl_prompt(
                    text_id=iid[:, 3:-5], ref_id=ref_ids[i][:, 3:-2],
                    ref_code=torch.as_tensor(voice_clone_prompt["ref_code"][i]).to(dev).clone(),
                    tts_pad_embed=pad, tts_eos_embed=eos, non_streaming_mode=non_streaming_mode)
                parts.append(icl_embed)
            elif non_streaming_mode:
                body = iid[:, 3:-5]
                parts.append(torch.cat((text(body), eos), dim=1) + emb_c(ids_t([[tc.codec_pad_id] * (body.shape[1] + 1)])))
                parts.append(pad + emb_c(ids_t([[tc.codec_bos_id]])))
                trail = pad
            else:
                parts.append(text(iid[:, 3:4]) + cod[:, -1:])
                trail = torch.cat((text(iid[:, 4:-5]), eos), dim=1)
            seqs.append(torch.cat(parts, dim=1).squeeze(0))
            trailing.append(trail.squeeze(0))
        # left-pad the batch (B is always 1 through the public API)
        L = max(s.shape[0] for s in seqs)
        H = seqs[0].shape[1]
        embeds = torch.zeros(len(seqs), L, H, dtype=seqs[0].dtype, device=dev)
        mask = torch.zeros(len(seqs), L, dtype=torch.long, device=dev)
        for b, s in enumerate(seqs):
            embeds[b, L - s.shape[0]:] = s
            mask[b, L - s.shape[0]:] = 1
        Tt = max(t.shape[0] for t in trailing)
        tth = pad.squeeze(0).expand(len(seqs), Tt, H).clone()
        for b, t in enumerate(trailing):
            tth[b, : t.shape[0]] = t
        return embeds, mask, tth, pad

    def _after_prepare(self, m, tie):
        if not self._warmed_up:
            self.warmup(tie.shape[1])
        m.talker.rope_deltas = None
        return m.talker, m.config.talker_config

    def _prepare_generation(self, text: str, ref_audio=None, ref_text: str = "", language: str = "English",
                            xvec_only: bool = False, non_streaming_mode: bool = False, append_silence: bool = True,
                            voice_clone_prompt=None, instruct: Optional[str] = None):
        input_ids = self.model._tokenize_texts([self.model._build_assistant_text(text)])
        instruct_ids = [None]
        if instruct:
            instruct_ids = [self.model._tokenize_texts([self.model._build_instruct_text(instruct)])[0]]
        vcp, ref_ids, using_icl = self._resolve_voice_clone_prompt(
            input_ids=input_ids, ref_audio=ref_audio, ref_text=ref_text, xvec_only=xvec_only,
            append_silence=append_silence, voice_clone_prompt=voice_clone_prompt)
        if instruct and not using_icl:
            logger.warning("Base-model instruct with x-vector-only voice cloning is experimental; prefer xvec_only=False.")
        m = self.model.model
        tie, tam, tth, tpe = self._build_talker_inputs_local(
            m=m, input_ids=input_ids, ref_ids=ref_ids, voice_clone_prompt=vcp,
            languages=[language] if language is not None else ["Auto"], speakers=None,
            non_streaming_mode=non_streaming_mode, instruct_ids=instruct_ids)
        talker, config = self._after_prepare(m, tie)
        ref_codes = None
        if using_icl and vcp.get("ref_code") and vcp["ref_code"][0] is not None:
            ref_codes = torch.as_tensor(vcp["ref_code"][0])
        return m, talker, config, tie, tam, tth, tpe, ref_codes

    def _prepare_generation_custom(self, text: str, language: str, speaker: Optional[str],
                                   instruct: Optional[str] = None, non_streaming_mode: bool = True):
        input_ids = self.model._tokenize_texts([self.model._build_assistant_text(text)])
        instruct_ids = [None if not instruct else
                        self.model._tokenize_texts([self.model._build_instruct_text(instruct)])[0]]
        m = self.model.model
        tie, tam, tth, tpe = self._build_talker_inputs_local(
            m=m, input_ids=input_ids, ref_ids=[None], voice_clone_prompt=None,
            languages=[language] if language is not None else ["Auto"], speakers=[speaker],
            non_streaming_mode=non_streaming_mode, instruct_ids=instruct_ids)
        talker, config = self._after_prepare(m, tie)
        return m, talker, config, tie, tam, tth, tpe

    # ---- shared back halves -------------------------------------------------------------------------------------
    def _run_full(self, m, talker, config, tie, tam, tth, tpe, ref_codes, gen_kwargs) -> Tuple[list, int]:
        from .generate import fast_generate
        codec_ids, timing = fast_generate(talker=talker, talker_input_embeds=tie, attention_mask=tam,
                                          trailing_text_hiddens=tth, tts_pad_embed=tpe, config=config,
                                          predictor_graph=self.predictor_graph, talker_graph=self.talker_graph,
                                          **gen_kwargs)
        if codec_ids is None:
            logger.warning("Generation returned no tokens")
            return [np.zeros(1, dtype=np.float32)], self.sample_rate
        # ICL: the reference codes go in front so the decoder has acoustic context (model.py:919-937)
        codes = torch.cat([ref_codes.to(codec_ids.device), codec_ids], dim=0) if ref_codes is not None else codec_ids
        ref_len = ref_codes.shape[0] if ref_codes is not None else 0
        tok = m.speech_tokenizer
        if ref_len > 0 and hasattr(tok, "decode_tensor") and hasattr(tok, "num_samples_total"):
            # the reference decodes ref + generated frames and cuts the reference's share (model.py:927-930); the HIP
            # decoder produces only that tail (bit-identical to the slice, fq3_codec_decode_tail)
            cut = int(ref_len / max(codes.shape[0], 1) * tok.num_samples_total(codes.shape[0]))
            pf = tok.prefix_for(ref_codes) if hasattr(tok, "prefix_for") else None          # the voice's cached front-end state
            out, sr = [_to_numpy(tok.decode_tensor(codes, cut, prefix=pf) if pf is not None else tok.decode_tensor(codes, cut))], tok.sample_rate
        else:
            audio_list, sr = tok.decode({"audio_codes": codes.unsqueeze(0)})
            out = []
            for a in audio_list:
                a = _to_numpy(a)
                if ref_len > 0:
                    a = a[int(ref_len / max(codes.shape[0], 1) * len(a)):]
                out.append(a)
        n = timing["steps"]
        total = timing["prefill_ms"] / 1000 + timing["decode_s"]
        logger.info("Generated %.2fs audio in %.2fs (%.1fms/step, RTF: %.2f)", n / 12.5, total, timing["ms_per_step"],
                    (n / 12.5) / total if total > 0 else 0)
        return out, sr

    def _vocoder_stream(self, tok):
        """The HIP stream every streaming vocoder of this model runs on (one codec workspace -> one stream), or None for a
        foreign tokenizer that is decoded synchronously."""
        use_side = torch.cuda.is_available() and hasattr(tok, "decode_tensor") and hasattr(tok, "num_samples_total")
        if not use_side:
            return None
        if getattr(self, "_voc_stream", None) is None:
            dev = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
            # vocoder_stream_priority (attribute, default None = the device default): HIP stream priority of the vocoder
            # stream, larger = lower.  Set it before the first streaming call.
            prio = getattr(self, "vocoder_stream_priority", None)
            share = getattr(self, "vocoder_cu_share", None)
            if share:
                # measurement switch (attribute, default None): the vocoder confined to a share of the CUs (fq3hip/streams.py::cu_masked_stream)
                from .streams import cu_masked_stream
                self._voc_stream = cu_masked_stream(dev, float(share))
                return self._voc_stream
            from .streams import concurrent_stream
            # a stream VERIFIED to execute beside the decode stream: one that shares its hardware queue would hold the first audio chunk
            # back until the next chunk's frames -- queued before it -- have run (fq3hip/streams.py)
            self._voc_stream = concurrent_stream(dev, prio)
        return self._voc_stream

    def streaming_vocoder(self, ref_codes, chunk_size: int) -> "StreamingVocoder":
        """A fresh windowing state for one utterance (used by the streaming entry points and by the batch server)."""
        tok = self.model.model.speech_tokenizer
        return StreamingVocoder(tok, ref_codes, chunk_size, self.device, self._vocoder_stream(tok))

    def _run_streaming(self, m, talker, config, tie, tam, tth, tpe, ref_codes, gen_kwargs, chunk_size: int,
                       parity_mode: bool = False):
        """model.py:1048-1137: the decode loop yields code chunks, ``StreamingVocoder`` turns each into the audio the
        reference's windowing would emit for it."""
        from .streaming import fast_generate_streaming, parity_generate_streaming
        tok = m.speech_tokenizer
        voc = StreamingVocoder(tok, ref_codes, chunk_size, self.device, self._vocoder_stream(tok))
        fn = parity_generate_streaming if parity_mode else fast_generate_streaming
        stream = fn(talker=talker, talker_input_embeds=tie, attention_mask=tam, trailing_text_hiddens=tth,
                    tts_pad_embed=tpe, config=config, predictor_graph=self.predictor_graph,
                    talker_graph=self.talker_graph, chunk_size=chunk_size, **gen_kwargs)
        for chunk, timing in stream:
            ev = timing.pop("codes_ready_event", None)
            new_audio, sr = voc.push(chunk, ev)
            yield new_audio, sr, timing

    @staticmethod
    def _gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p, do_sample, repetition_penalty):
        return dict(max_new_tokens=max_new_tokens, min_new_tokens=min_new_tokens, temperature=temperature,
                    top_k=top_k, top_p=top_p, do_sample=do_sample, repetition_penalty=repetition_penalty)

    # ---- public generation entry points ----------------------------------------------------------------------------
    @torch.inference_mode()
    def generate_voice_clone(self, text: str, language: str, ref_audio: Optional[Union[str, Path]] = None,
                             ref_text: str = "", max_new_tokens: int = 2048, min_new_tokens: int = 2,
                             temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0, do_sample: bool = True,
                             repetition_penalty: float = 1.05, xvec_only: bool = False,
                             non_streaming_mode: Optional[bool] = None, append_silence: bool = True,
                             instruct: Optional[str] = None, ref_spk: Optional[Union[str, Path]] = None,
                             ref_rvq: Optional[Union[str, Path]] = None, ref_spk_emb: Optional[np.ndarray] = None,
                             ref_codes: Optional[np.ndarray] = None,
                             voice_clone_prompt: Optional[Union[Dict[str, Any], List[Any]]] = None) -> Tuple[list, int]:
        """Voice cloning; returns ``([np.float32 waveform], sample_rate)``."""
        self._reject_ggml_cached_reference_args(ref_spk=ref_spk, ref_rvq=ref_rvq, ref_spk_emb=ref_spk_emb,
                                                ref_codes=ref_codes)
        nsm = self._resolve_non_streaming_mode(non_streaming_mode, default=False)
        m, talker, config, tie, tam, tth, tpe, rc = self._prepare_generation(
            text=text, language=language, ref_audio=ref_audio, ref_text=ref_text, xvec_only=xvec_only,
            non_streaming_mode=nsm, append_silence=append_silence, voice_clone_prompt=voice_clone_prompt,
            instruct=instruct)
        return self._run_full(m, talker, config, tie, tam, tth, tpe, rc,
                              self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p, do_sample,
                                               repetition_penalty))

    # ---- batched generation (extension: the reference has no multi-utterance entry point) ---------------------------------
    def _batch_decoder(self, lanes: int, staging: Optional[int] = None):
        """Lazily builds ``lanes`` decode contexts over this model's single weight replica, ``staging`` spare contexts
        (default: as many as lanes) that the next requests are prefilled into while the batch decodes, the KV block pool they
        all draw from, and the scheduler on top.

        ``self.batch_kv_blocks`` (attribute, default ``None``): size of that pool in 64-key blocks.  ``None`` = enough for every
        lane AND every spare context at ``max_seq_len`` (what static caches would reserve; the pool then never runs short).  A
        server that knows its traffic sets less -- e.g. ``(lanes + staging) * ceil((prompt + max_new_tokens + 1) / 64)`` --
        and a request the pool cannot hold yet waits until finished lanes give blocks back.  NOTE: a request reserves its WORST case up
        front (``prompt + max_new_tokens + 1`` key slots, capped at ``max_seq_len``: a short pool must show before any work is queued),
        so with the default ``max_new_tokens=2048`` at ``max_seq_len=2048`` every request takes a whole context's blocks and a reduced pool
        merely serialises requests -- size ``batch_kv_blocks`` from the ``max_new_tokens`` the callers really pass."""
        from .batching import BatchDecoder
        from .engine import Fq3Engine, Fq3KvPool
        from .batching import MAX_LANES
        lanes = max(1, min(int(lanes), MAX_LANES))
        staging = lanes if staging is None else max(0, int(staging))
        first = self.talker_graph.engine
        full = (lanes + staging) * Fq3KvPool.blocks_for(first.max_seq_len)
        want = getattr(self, "batch_kv_blocks", None)
        blocks = full if want is None else max(Fq3KvPool.blocks_for(first.max_seq_len), min(int(want), full))
        # batch_groups (attribute, default None = the library's choice: one chain): fq3_batch_set_option("groups"), a measurement switch
        groups = getattr(self, "batch_groups", None)
        cached = getattr(self, "_batch_cache", None)
        if cached is not None and cached[0] == (lanes, staging, blocks, groups):
            la = getattr(self, "batch_lookahead", None)
            cached[1].lookahead = (int(la) if la is not None else 1) if hasattr(cached[1].batch, "poll_async") else 0
            # the lanes follow this model's CURRENT predictor policy (it is copied into the loop state when a lane is armed)
            pg = self.predictor_graph
            cached[1].set_predictor_policy(do_sample=pg.do_sample, top_k=pg.top_k, top_p=pg.top_p, temperature=pg.temperature)
            return cached[1]
        self._batch_cache = None                    # a differently shaped scheduler: its contexts and pool go first
        pool = Fq3KvPool(first.cfg, blocks, device=str(first.device), dtype=first.dtype)
        mk = lambda: Fq3Engine(first.cfg, first.weights, device=str(first.device), dtype=first.dtype,
                               max_seq_len=first.max_seq_len, max_frames=first.max_frames, share=first, pool=pool)
        engines = [mk() for _ in range(lanes)]
        pg = self.predictor_graph
        dec = BatchDecoder(engines, predictor_policy=dict(do_sample=pg.do_sample, top_k=pg.top_k, top_p=pg.top_p,
                                                          temperature=pg.temperature),
                           staging=[mk() for _ in range(staging)])
        dec.kv_pool = pool
        # batch_lookahead (attribute, default None = the scheduler's default, 1: the next batch of frames is queued before the previous
        # batch's poll is read; 0 = wait for every batch right after queuing it) -- see BatchDecoder.lookahead
        la = getattr(self, "batch_lookahead", None)
        if la is not None:
            dec.lookahead = int(la) if hasattr(dec.batch, "poll_async") else 0
        if groups is not None and hasattr(dec.batch, "set_option"):
            dec.batch.set_option("groups", int(groups))
            dec.n_groups = int(groups)
        if torch.cuda.is_available():
            tok = self.model.model.speech_tokenizer
            dec.beside = [st for st in (self._vocoder_stream(tok),) if st is not None]
        self._batch_cache = ((lanes, staging, blocks, groups), dec)
        return dec

    def _side_vocoder(self):
        """Vocoder for finished utterances of a batched run: decodes on its own HIP stream (the lock-step decode of the
        remaining / next utterances goes on meanwhile) and copies the waveform to pinned host memory asynchronously."""
        tok = self.model.model.speech_tokenizer
        if getattr(self, "_side_voc_stream", None) is None and torch.cuda.is_available() and hasattr(tok, "decode_tensor"):
            from .streams import concurrent_stream
            # ONE vocoder stream per model (probed once, fq3hip/streams.py): hardware queues are few, and the scheduler's prefill
            # stream and the lane groups' stream have to stay clear of it
            self._side_voc_stream = self._vocoder_stream(tok) or concurrent_stream(self.device)
        return _SideVocoder(tok, self.device, getattr(self, "_side_voc_stream", None))

    def _batch_feed(self, prepared, gen_kwargs, lanes: int, meta: dict, first_wave: Optional[int] = None):
        """``prepared``: an iterable (usually a generator: the prompt of utterance i is built when the scheduler asks for it) of
        ``(talker, config, tie, tam, tth, tpe, ref_codes | None)``.  Returns ``(head, source)`` for ``BatchDecoder.run``: the first
        ``lanes`` requests up front -- the first wave starts decoding before the later prompts exist -- and a ``source`` callback
        that prepares the rest one at a time at frame boundaries.  ``meta[i]`` receives the entry's ``ref_codes``."""
        from .batching import BatchRequest
        it = enumerate(prepared)

        def pull():
            nxt = next(it, None)
            if nxt is None:
                return None
            i, (talker, config, tie, tam, tth, tpe, rc) = nxt
            meta[i] = rc
            return BatchRequest(i, talker, tie, tam, tth, tpe, config, dict(gen_kwargs))

        # (only the first SLICE of the first wave is built here: the scheduler pulls the rest from `source` in slices and queues every
        # slice's packed prefill behind the one before, so prompt builds and prefills overlap -- BatchDecoder.run, "first wave, pipelined")
        head = []
        wave = max(1, int(lanes) if first_wave is None else min(int(lanes), int(first_wave)))
        for _ in range(min(wave, 16)):
            r = pull()
            if r is None:
                break
            head.append(r)
        return head, pull

    def _run_batch_full(self, prepared, gen_kwargs, lanes: int, count: int) -> List[Tuple[list, int]]:
        """The lock-step decode of ``count`` prepared utterances (see :meth:`_batch_feed`) through ``lanes`` lanes, every finished
        utterance vocoded on the side stream (the reference's share of an ICL waveform is cut by not producing it,
        model.py:927-930).  One ``([waveform], sample_rate)`` per entry, in input order."""
        meta: dict = {}
        dec = self._batch_decoder(lanes)
        # batch_first_wave (attribute): requests prepared + prefilled + armed before the first frame is queued (BatchDecoder.first_wave);
        # None = one per lane
        dec.first_wave = getattr(self, "batch_first_wave", None)
        head, source = self._batch_feed(prepared, gen_kwargs, len(dec.lanes), meta, dec.first_wave)
        out: List[Optional[Tuple[list, int]]] = [None] * count
        voc = self._side_vocoder()
        # batch_vocode_every (attribute; default 0 = vocode an utterance when it has ended; N > 0 = produce the waveform in slices every N
        # frames while the utterance still decodes -- exact, see _SideVocoder.inc_add -- so that what is left at the end is the last slice
        # only).  Measured on MI355X (profiles/r04_batch_e2e_vocode_every_*.txt): the slices contend with the latency-bound lock-step
        # frames for more than they save at the end -- 64 lanes x 128 utterances, 0.6B: 561x at 0, 537x at 64, 552x at 100; 1.7B: 458x vs
        # 443x -- so it is OFF by default; a server whose lanes finish at different times has nothing to gain from it either.
        inc = int(getattr(self, "batch_vocode_every", 0) or 0)
        if inc > 0 and voc.async_ok and hasattr(voc.tok, "decode_tensor_batch") and hasattr(voc.tok, "num_samples_total"):
            max_new = min(int(gen_kwargs.get("max_new_tokens", 2048)), int(self.talker_graph.engine.max_frames))
            for rid, codes, info in dec.run(head, source=source, chunk_frames=inc):
                more = int(getattr(dec, "more_in_poll", 0))
                voc.inc_add(rid, codes, meta.get(rid), info.pop("codes_ready_event", None), bool(info.get("is_final")), more, max_new)
            for rid, a in voc.inc_collect():
                out[rid] = ([a], voc.sample_rate)
            return [o if o is not None else ([np.zeros(1, dtype=np.float32)], self.sample_rate) for o in out]
        for rid, codec_ids, timing in dec.run(head, source=source):
            more = int(getattr(dec, "more_in_poll", 0))
            if codec_ids is None:
                out[rid] = ([np.zeros(1, dtype=np.float32)], self.sample_rate)
                if more <= 0:
                    voc.add_flush()
                continue
            rc = meta[rid]
            codes = torch.cat([rc.to(codec_ids.device), codec_ids], dim=0) if rc is not None else codec_ids
            # side stream; the next frames of the other lanes are not held up.  Utterances that finished in the same poll are vocoded
            # together (equal lengths: one batched launch set)
            voc.add(rid, codes, ref_len=rc.shape[0] if rc is not None else 0, more=more, ref=rc)
        voc.add_flush()
        for rid, a in voc.collect():
            out[rid] = ([a], voc.sample_rate)
        return out

    def _run_batch_streaming(self, prepared, gen_kwargs, chunk_size: int, lanes: int):
        """Streaming form of :meth:`_run_batch_full`: yields ``(index, audio_chunk, sample_rate, timing)`` for every
        ``chunk_size`` frames of any utterance -- per utterance exactly the chunks the single-utterance streaming entry point
        would produce (same windowing state machine).  ``timing``: ``chunk_index``, ``total_steps_so_far``, ``is_final``."""
        meta: dict = {}
        vocs, n_chunks = {}, {}
        dec = self._batch_decoder(lanes)
        # ``batch_first_wave_streaming`` (attribute, default None = one request per lane): requests prepared and prefilled in front of the
        # first frame, the others joining at the following frame boundaries.  MEASURED (profiles/r05_ttfa_first_wave.txt): a smaller
        # first wave does not buy its members a lower latency -- the prefills of the followers run beside their first frames and
        # stretch them (128 lanes: p25 305 ms / p50 395 ms at a wave of 32 against 340 ms for everybody at once) -- so the default stays.
        dec.first_wave = getattr(self, "batch_first_wave_streaming", None)
        head, source = self._batch_feed(prepared, gen_kwargs, len(dec.lanes), meta, dec.first_wave)
        tok = self.model.model.speech_tokenizer
        side = self._vocoder_stream(tok)
        batched = side is not None and hasattr(tok, "decode_tensor_batch")
        jobs: list = []                                  # the events of one poll, in order: [rid, meta, audio | None, (codes_in, first, ev) | None]

        # chunks of different utterances that need the SAME decode shape (the first chunks of lanes that started together: reference + 8
        # frames in, the last 8 frames' samples out) go through one batched launch set on the vocoder stream, STREAM_GROUP at a time.
        # Round 6: a group is LAUNCHED as soon as it is full -- while the host still prepares the next group's inputs -- every group's
        # waveform goes to pinned host memory behind its launch set, and the groups are handed out one by one as their copies land: a
        # wave of 128 first chunks (four groups of 32) used to be prepared as a whole, launched group by group with a host wait after
        # each, and reached the caller all at once after the last group.  Per utterance the order of the events is unchanged: groups are
        # handed out in launch order, a launch takes every open class in order of first appearance, markers without audio come last.
        open_classes: Dict[Any, list] = {}
        queued: list = []                               # (jobs of the group, pinned waveforms, event, device waveforms kept alive)

        def launch(key, part):
            _T, first, pf_len = key
            for j in part:
                if j[3][2] is not None:
                    side.wait_event(j[3][2])
                else:
                    side.wait_stream(torch.cuda.current_stream(torch.device(self.device)))
            with torch.cuda.stream(side):
                pfs = [j[3][3] for j in part] if pf_len > 0 else None      # (phase 1 behind ICL references: the voices' cached front-end states)
                if len(part) == 1:
                    wav = (tok.decode_tensor(part[0][3][0], first, prefix=pfs[0]) if pfs else tok.decode_tensor(part[0][3][0], first)).reshape(1, -1)
                else:
                    stacked = torch.stack([j[3][0] for j in part])
                    wav = tok.decode_tensor_batch(stacked, first, prefixes=pfs) if pfs else tok.decode_tensor_batch(stacked, first)
                wav = wav.float()
                host = torch.empty(wav.shape, dtype=wav.dtype, pin_memory=True)
                host.copy_(wav, non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(side)
            queued.append((part, host, landed, wav))

        def launch_open():
            for key in list(open_classes):
                members = open_classes.pop(key)
                for i in range(0, len(members), _SideVocoder.STREAM_GROUP):
                    launch(key, members[i:i + _SideVocoder.STREAM_GROUP])

        def add_job(j):
            jobs.append(j)
            if j[3] is not None:
                pf = j[3][3] if len(j[3]) > 3 else None
                key = (int(j[3][0].shape[0]), int(j[3][1]), pf.ref_len if pf is not None else 0)
                open_classes.setdefault(key, []).append(j)
                if len(open_classes[key]) >= _SideVocoder.STREAM_GROUP:
                    launch_open()

        def flush():
            launch_open()
            plain = [j for j in jobs if j[3] is None]
            jobs[:] = []
            groups, queued[:] = list(queued), []
            for part, host, landed, _wav in groups:
                landed.synchronize()
                arr = host.numpy()
                for k, j in enumerate(part):
                    j[2] = arr[k]
                yield from part
            yield from plain

        for rid, codes, info in dec.run(head, source=source, chunk_frames=chunk_size):
            if rid not in vocs:
                vocs[rid], n_chunks[rid] = self.streaming_vocoder(meta[rid], chunk_size), 0
            ev = info.pop("codes_ready_event", None)
            final = bool(info.get("is_final"))
            more = int(getattr(dec, "more_in_poll", 0))
            out_meta = dict(chunk_index=n_chunks[rid], total_steps_so_far=int(info.get("total_steps_so_far", 0)),
                            is_final=final, chunk_steps=0 if codes is None else int(codes.shape[0]))
            if codes is not None and codes.shape[0] > 0:
                if batched:
                    inp, first = vocs[rid].prepare(codes, ev)
                    add_job([rid, out_meta, None, (inp, first, ev, vocs[rid].last_prefix)])
                else:
                    audio, _sr = vocs[rid].push(codes, ev)
                    add_job([rid, out_meta, audio, None])
                n_chunks[rid] += 1
            elif final:
                add_job([rid, out_meta, np.zeros(1 if codes is None else 0, dtype=np.float32), None])
                n_chunks[rid] += 1
            if more <= 0:
                for r, m, audio, _job in flush():
                    yield r, audio, self.sample_rate, m
        for r, m, audio, _job in flush():
            yield r, audio, self.sample_rate, m

    @staticmethod
    def _per_text(value, n: int, what: str) -> list:
        """One value for all texts, or one per text."""
        if isinstance(value, (list, tuple)):
            if len(value) != n:
                raise ValueError(f"{what} must be one value or one per text")
            return list(value)
        return [value] * n

    def _prepare_clone_batch(self, texts, language, ref_audio, ref_text, xvec_only, nsm, append_silence, instruct, voice_clone_prompt):
        langs = self._per_text(language, len(texts), "language")            # argument errors surface before anything is decoded

        def gen():
            for text, lang in zip(texts, langs):
                _m, talker, config, tie, tam, tth, tpe, rc = self._prepare_generation(
                    text=text, language=lang, ref_audio=ref_audio, ref_text=ref_text, xvec_only=xvec_only,
                    non_streaming_mode=nsm, append_silence=append_silence, voice_clone_prompt=voice_clone_prompt,
                    instruct=instruct)
                yield talker, config, tie, tam, tth, tpe, rc
        return gen()

    @torch.inference_mode()
    def generate_voice_clone_batch(self, texts: List[str], language: Union[str, List[str]] = "English",
                                   ref_audio: Optional[Union[str, Path]] = None, ref_text: str = "",
                                   max_new_tokens: int = 2048, min_new_tokens: int = 2, temperature: float = 0.9,
                                   top_k: int = 50, top_p: float = 1.0, do_sample: bool = True,
                                   repetition_penalty: float = 1.05, xvec_only: bool = False,
                                   non_streaming_mode: Optional[bool] = None, append_silence: bool = True,
                                   instruct: Optional[str] = None,
                                   voice_clone_prompt: Optional[Union[Dict[str, Any], List[Any]]] = None,
                                   lanes: int = 8) -> List[Tuple[list, int]]:
        """Voice cloning of several texts with one voice: up to ``lanes`` (<= 128) utterances decode in lock-step over
        one pass of the weights per frame (``fq3_batch_*``), finished lanes are refilled from the queue.  Returns one
        ``([np.float32 waveform], sample_rate)`` per text, in input order; each utterance follows exactly the
        single-utterance semantics of :meth:`generate_voice_clone`, nucleus sampling (``top_p < 1``) included."""
        nsm = self._resolve_non_streaming_mode(non_streaming_mode, default=False)
        prepared = self._prepare_clone_batch(texts, language, ref_audio, ref_text, xvec_only, nsm, append_silence, instruct,
                                             voice_clone_prompt)
        return self._run_batch_full(prepared, self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p, do_sample,
                                                               repetition_penalty), lanes, len(texts))

    @torch.inference_mode()
    def generate_voice_clone_batch_streaming(self, texts: List[str], language: Union[str, List[str]] = "English",
                                             ref_audio: Optional[Union[str, Path]] = None, ref_text: str = "",
                                             max_new_tokens: int = 2048, min_new_tokens: int = 2, temperature: float = 0.9,
                                             top_k: int = 50, top_p: float = 1.0, do_sample: bool = True,
                                             repetition_penalty: float = 1.05, chunk_size: int = 12, xvec_only: bool = False,
                                             non_streaming_mode: Optional[bool] = None, append_silence: bool = True,
                                             instruct: Optional[str] = None,
                                             voice_clone_prompt: Optional[Union[Dict[str, Any], List[Any]]] = None,
                                             lanes: int = 8) -> Generator[Tuple[int, np.ndarray, int, dict], None, None]:
        """Streaming AND batched: the texts decode in lock-step lanes (as :meth:`generate_voice_clone_batch`) and every
        ``chunk_size`` frames of any utterance yield ``(text_index, audio_chunk, sample_rate, timing)`` -- per utterance
        exactly the chunks :meth:`generate_voice_clone_streaming` would produce for it (same windowing state machine).
        ``timing``: ``chunk_index``, ``total_steps_so_far``, ``is_final``."""
        nsm = self._resolve_non_streaming_mode(non_streaming_mode, default=False)
        prepared = self._prepare_clone_batch(texts, language, ref_audio, ref_text, xvec_only, nsm, append_silence, instruct,
                                             voice_clone_prompt)
        yield from self._run_batch_streaming(prepared, self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p,
                                                                         do_sample, repetition_penalty), chunk_size, lanes)

    def _prepare_custom_batch(self, texts, speaker, language, instruct, non_streaming_mode):
        n = len(texts)
        spks, langs, inss = self._per_text(speaker, n, "speaker"), self._per_text(language, n, "language"), self._per_text(instruct, n, "instruct")
        if self.model.model.tts_model_type != "custom_voice":
            raise ValueError("Loaded model does not support custom voice generation")
        self.model._validate_languages(langs)                                # every argument error before anything is decoded
        self.model._validate_speakers(spks)

        def gen():
            for text, spk, lang, ins in zip(texts, spks, langs, inss):
                _m, talker, config, tie, tam, tth, tpe = self._custom_prepare(text, spk, lang, ins, non_streaming_mode)
                yield talker, config, tie, tam, tth, tpe, None
        return gen()

    @torch.inference_mode()
    def generate_custom_voice_batch(self, texts: List[str], speaker: Union[str, List[str]], language: Union[str, List[str]] = "English",
                                    instruct: Optional[Union[str, List[Optional[str]]]] = None,
                                    non_streaming_mode: Optional[bool] = None, max_new_tokens: int = 2048, min_new_tokens: int = 2,
                                    temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0, do_sample: bool = True,
                                    repetition_penalty: float = 1.05, lanes: int = 16) -> List[Tuple[list, int]]:
        """CustomVoice for several texts (BASELINE configs[3]: many concurrent utterances of a CustomVoice model): every
        utterance is prepared exactly like :meth:`generate_custom_voice` (speaker-id prompt, model.py:1139-1326; ``speaker`` /
        ``language`` / ``instruct`` may be one value or one per text) and up to ``lanes`` of them decode in lock-step over one pass of
        the weights per frame.  Returns one ``([waveform], sample_rate)`` per text, in input order."""
        prepared = self._prepare_custom_batch(texts, speaker, language, instruct, non_streaming_mode)
        return self._run_batch_full(prepared, self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p, do_sample,
                                                               repetition_penalty), lanes, len(texts))

    @torch.inference_mode()
    def generate_custom_voice_batch_streaming(self, texts: List[str], speaker: Union[str, List[str]],
                                              language: Union[str, List[str]] = "English",
                                              instruct: Optional[Union[str, List[Optional[str]]]] = None,
                                              non_streaming_mode: Optional[bool] = None, max_new_tokens: int = 2048,
                                              min_new_tokens: int = 2, temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0,
                                              do_sample: bool = True, repetition_penalty: float = 1.05, chunk_size: int = 12,
                                              lanes: int = 16) -> Generator[Tuple[int, np.ndarray, int, dict], None, None]:
        """Streaming form of :meth:`generate_custom_voice_batch`: ``(text_index, audio_chunk, sample_rate, timing)`` per chunk."""
        prepared = self._prepare_custom_batch(texts, speaker, language, instruct, non_streaming_mode)
        yield from self._run_batch_streaming(prepared, self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p,
                                                                         do_sample, repetition_penalty), chunk_size, lanes)

    @torch.inference_mode()
    def generate_voice_design_batch(self, texts: List[str], instruct: Union[str, List[str]], language: Union[str, List[str]] = "English",
                                    non_streaming_mode: Optional[bool] = None, max_new_tokens: int = 2048, min_new_tokens: int = 2,
                                    temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0, do_sample: bool = True,
                                    repetition_penalty: float = 1.05, lanes: int = 16) -> List[Tuple[list, int]]:
        """VoiceDesign for several texts in lock-step lanes; every utterance prepared like :meth:`generate_voice_design`."""
        return self._run_batch_full(self._prepare_design_batch(texts, instruct, language, non_streaming_mode),
                                    self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p, do_sample, repetition_penalty),
                                    lanes, len(texts))

    def _prepare_design_batch(self, texts, instruct, language, non_streaming_mode):
        n = len(texts)
        inss, langs = self._per_text(instruct, n, "instruct"), self._per_text(language, n, "language")
        if self.model.model.tts_model_type != "voice_design":
            raise ValueError("Loaded model does not support voice design generation")
        self.model._validate_languages(langs)

        def gen():
            for text, ins, lang in zip(texts, inss, langs):
                _m, talker, config, tie, tam, tth, tpe = self._design_prepare(text, ins, lang, non_streaming_mode)
                yield talker, config, tie, tam, tth, tpe, None
        return gen()

    @torch.inference_mode()
    def generate_voice_design_batch_streaming(self, texts: List[str], instruct: Union[str, List[str]],
                                              language: Union[str, List[str]] = "English", non_streaming_mode: Optional[bool] = None,
                                              max_new_tokens: int = 2048, min_new_tokens: int = 2, temperature: float = 0.9,
                                              top_k: int = 50, top_p: float = 1.0, do_sample: bool = True, repetition_penalty: float = 1.05,
                                              chunk_size: int = 12, lanes: int = 16) -> Generator[Tuple[int, np.ndarray, int, dict], None, None]:
        """Streaming form of :meth:`generate_voice_design_batch`: ``(text_index, audio_chunk, sample_rate, timing)`` per chunk."""
        prepared = self._prepare_design_batch(texts, instruct, language, non_streaming_mode)
        yield from self._run_batch_streaming(prepared, self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p,
                                                                         do_sample, repetition_penalty), chunk_size, lanes)

    @torch.inference_mode()
    def generate_voice_clone_streaming(self, text: str, language: str, ref_audio: Optional[Union[str, Path]] = None,
                                       ref_text: str = "", max_new_tokens: int = 2048, min_new_tokens: int = 2,
                                       temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0,
                                       do_sample: bool = True, repetition_penalty: float = 1.05, chunk_size: int = 12,
                                       xvec_only: bool = False, non_streaming_mode: Optional[bool] = None,
                                       append_silence: bool = True, parity_mode: bool = False,
                                       instruct: Optional[str] = None, ref_spk: Optional[Union[str, Path]] = None,
                                       ref_rvq: Optional[Union[str, Path]] = None,
                                       ref_spk_emb: Optional[np.ndarray] = None, ref_codes: Optional[np.ndarray] = None,
                                       voice_clone_prompt: Optional[Union[Dict[str, Any], List[Any]]] = None,
                                       ) -> Generator[Tuple[np.ndarray, int, dict], None, None]:
        """Yields ``(audio_chunk, sample_rate, timing)`` every ``chunk_size`` codec frames."""
        self._reject_ggml_cached_reference_args(ref_spk=ref_spk, ref_rvq=ref_rvq, ref_spk_emb=ref_spk_emb,
                                                ref_codes=ref_codes)
        nsm = self._resolve_non_streaming_mode(non_streaming_mode, default=False)
        m, talker, config, tie, tam, tth, tpe, rc = self._prepare_generation(
            text=text, language=language, ref_audio=ref_audio, ref_text=ref_text, xvec_only=xvec_only,
            non_streaming_mode=nsm, append_silence=append_silence, voice_clone_prompt=voice_clone_prompt,
            instruct=instruct)
        yield from self._run_streaming(m, talker, config, tie, tam, tth, tpe, rc,
                                       self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p,
                                                        do_sample, repetition_penalty), chunk_size, parity_mode)

    def _custom_prepare(self, text, speaker, language, instruct, non_streaming_mode):
        if self.model.model.tts_model_type != "custom_voice":
            raise ValueError("Loaded model does not support custom voice generation")
        self.model._validate_languages([language])
        self.model._validate_speakers([speaker])
        nsm = self._resolve_non_streaming_mode(non_streaming_mode, default=True)
        if self.model.model.tts_model_size in "0b6":        # 0.6B CustomVoice ignores instruct (model.py:1166-1167)
            instruct = None
        return self._prepare_generation_custom(text=text, language=language, speaker=speaker, instruct=instruct,
                                               non_streaming_mode=nsm)

    @torch.inference_mode()
    def generate_custom_voice(self, text: str, speaker: str, language: str, instruct: Optional[str] = None,
                              non_streaming_mode: Optional[bool] = None, max_new_tokens: int = 2048,
                              min_new_tokens: int = 2, temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0,
                              do_sample: bool = True, repetition_penalty: float = 1.05) -> Tuple[list, int]:
        m, talker, config, tie, tam, tth, tpe = self._custom_prepare(text, speaker, language, instruct, non_streaming_mode)
        return self._run_full(m, talker, config, tie, tam, tth, tpe, None,
                              self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p, do_sample,
                                               repetition_penalty))

    @torch.inference_mode()
    def generate_custom_voice_streaming(self, text: str, speaker: str, language: str, instruct: Optional[str] = None,
                                        non_streaming_mode: Optional[bool] = None, max_new_tokens: int = 2048,
                                        min_new_tokens: int = 2, temperature: float = 0.9, top_k: int = 50,
                                        top_p: float = 1.0, do_sample: bool = True, repetition_penalty: float = 1.05,
                                        chunk_size: int = 12) -> Generator[Tuple[np.ndarray, int, dict], None, None]:
        m, talker, config, tie, tam, tth, tpe = self._custom_prepare(text, speaker, language, instruct, non_streaming_mode)
        yield from self._run_streaming(m, talker, config, tie, tam, tth, tpe, None,
                                       self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p,
                                                        do_sample, repetition_penalty), chunk_size)

    def _design_prepare(self, text, instruct, language, non_streaming_mode):
        if self.model.model.tts_model_type != "voice_design":
            raise ValueError("Loaded model does not support voice design generation")
        self.model._validate_languages([language])
        nsm = self._resolve_non_streaming_mode(non_streaming_mode, default=True)
        return self._prepare_generation_custom(text=text, language=language, speaker=None, instruct=instruct,
                                               non_streaming_mode=nsm)

    @torch.inference_mode()
    def generate_voice_design(self, text: str, instruct: str, language: str, non_streaming_mode: Optional[bool] = None,
                              max_new_tokens: int = 2048, min_new_tokens: int = 2, temperature: float = 0.9,
                              top_k: int = 50, top_p: float = 1.0, do_sample: bool = True,
                              repetition_penalty: float = 1.05) -> Tuple[list, int]:
        m, talker, config, tie, tam, tth, tpe = self._design_prepare(text, instruct, language, non_streaming_mode)
        return self._run_full(m, talker, config, tie, tam, tth, tpe, None,
                              self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p, do_sample,
                                               repetition_penalty))

    @torch.inference_mode()
    def generate_voice_design_streaming(self, text: str, instruct: str, language: str,
                                        non_streaming_mode: Optional[bool] = None, max_new_tokens: int = 2048,
                                        min_new_tokens: int = 2, temperature: float = 0.9, top_k: int = 50,
                                        top_p: float = 1.0, do_sample: bool = True, repetition_penalty: float = 1.05,
                                        chunk_size: int = 12) -> Generator[Tuple[np.ndarray, int, dict], None, None]:
        m, talker, config, tie, tam, tth, tpe = self._design_prepare(text, instruct, language, non_streaming_mode)
        yield from self._run_streaming(m, talker, config, tie, tam, tth, tpe, None,
                                       self._gen_kwargs(max_new_tokens, min_new_tokens, temperature, top_k, top_p,
                                                        do_sample, repetition_penalty), chunk_size)
