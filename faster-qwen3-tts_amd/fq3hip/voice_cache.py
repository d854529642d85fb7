"""On-disk voice-reference cache: ``<key>.spk`` / ``<key>.rvq`` / ``<key>.json``.

The reference keeps such a cache for its GGML backend (``faster_qwen3_tts/ggml_backend.py:403-471``: key = SHA-256 of a
canonical metadata JSON that contains the audio's SHA-256, three files per entry, written through temporaries and
``rename``).  The payload encodings there belong to the external ``libqwen`` runtime; here they are defined by this file:

    <key>.spk   little-endian float32[H]            speaker embedding (x-vector)
    <key>.rvq   b"FQ3RVQ1\\0" + int32 T + int32 G + little-endian int16[T, G]    reference codec frames (ICL), G = 16
    <key>.json  the metadata the key was derived from (+ "ref_text")

An entry is exactly one ``voice_clone_prompt`` item (what upstream ``create_voice_clone_prompt`` returns,
reference ``model.py:430-451``).  Entries are written through by ``FasterQwen3TTS`` the first time a reference clip is analysed
(the HIP analysers, ``fq3hip/refenc.py``; or ``export_voice_clone_prompt`` below over an upstream prompt item) and every later
``generate_voice_clone(ref_audio=...)`` call -- in this or another process -- is served from the cache.  The key's metadata
carries the entry's *mode* (``"xvec"`` = speaker embedding only, ``"icl"`` = embedding + reference codes), so an x-vector-only
entry never answers an ICL request.
"""
from __future__ import annotations

import hashlib
import json
import os
import struct
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import numpy as np

CACHE_VERSION = 2          # 2: the key metadata carries the entry's mode
_RVQ_MAGIC = b"FQ3RVQ1\0"


def cache_key(ref_audio_24k: np.ndarray, *, append_silence: bool, model_identity: str, mode: str = "icl") -> Tuple[str, dict]:
    """Same construction as the reference (``ggml_backend.py:403-416``): metadata dict -> canonical JSON -> SHA-256.
    ``mode``: ``"icl"`` (speaker embedding + reference codes) or ``"xvec"`` (speaker embedding only)."""
    if mode not in ("icl", "xvec"):
        raise ValueError("mode must be 'icl' or 'xvec'")
    audio = np.ascontiguousarray(np.asarray(ref_audio_24k, dtype=np.float32))
    meta = {"version": CACHE_VERSION, "model_identity": str(model_identity), "sample_rate": 24000, "dtype": "float32",
            "n_samples": int(audio.shape[0]), "append_silence": bool(append_silence), "mode": mode,
            "audio_sha256": hashlib.sha256(audio.tobytes()).hexdigest()}
    payload = json.dumps(meta, sort_keys=True, separators=(",", ":")).encode("utf-8")
    return hashlib.sha256(payload).hexdigest(), meta


class VoiceRefCache:
    def __init__(self, directory):
        self.dir = Path(directory)

    def paths(self, key: str) -> Tuple[Path, Path, Path]:
        base = self.dir / key
        return base.with_suffix(".spk"), base.with_suffix(".rvq"), base.with_suffix(".json")

    def save(self, key: str, metadata: dict, spk_embedding, ref_code=None, ref_text: str = "") -> None:
        """Atomic (temporaries + rename, ``ggml_backend.py:448-465``)."""
        self.dir.mkdir(parents=True, exist_ok=True)
        spk_p, rvq_p, meta_p = self.paths(key)
        tmp = self.dir / f".{key}.{os.getpid()}"
        spk = np.asarray(_to_numpy(spk_embedding), dtype="<f4").reshape(-1)
        tmp.with_suffix(".spk").write_bytes(spk.tobytes())
        have_rvq = ref_code is not None
        if have_rvq:
            codes = np.asarray(_to_numpy(ref_code)).astype("<i2")
            if codes.ndim != 2:
                raise ValueError("ref_code must be [T, num_code_groups]")
            tmp.with_suffix(".rvq").write_bytes(_RVQ_MAGIC + struct.pack("<ii", *codes.shape) + codes.tobytes())
        tmp.with_suffix(".json").write_text(json.dumps(dict(metadata, ref_text=ref_text, has_rvq=have_rvq), sort_keys=True))
        tmp.with_suffix(".spk").replace(spk_p)
        if have_rvq:
            tmp.with_suffix(".rvq").replace(rvq_p)
        tmp.with_suffix(".json").replace(meta_p)

    def load(self, key: str, metadata: Optional[dict] = None) -> Optional[Dict[str, Any]]:
        """-> dict(ref_spk_embedding float32[H], ref_code int64[T, G] | None, ref_text) or None on a miss / stale entry."""
        spk_p, rvq_p, meta_p = self.paths(key)
        if not (spk_p.is_file() and meta_p.is_file()):
            return None
        try:
            meta = json.loads(meta_p.read_text())
            ref_text, has_rvq = meta.pop("ref_text", ""), meta.pop("has_rvq", rvq_p.is_file())
            if metadata is not None and meta != metadata:
                return None
            spk = np.frombuffer(spk_p.read_bytes(), dtype="<f4").copy()
            codes = None
            if has_rvq:
                raw = rvq_p.read_bytes()
                if raw[:8] != _RVQ_MAGIC:
                    return None
                T, G = struct.unpack("<ii", raw[8:16])
                codes = np.frombuffer(raw[16:], dtype="<i2").reshape(T, G).astype(np.int64)
            return dict(ref_spk_embedding=spk, ref_code=codes, ref_text=ref_text)
        except Exception:
            return None


def _to_numpy(x):
    if hasattr(x, "detach"):
        x = x.detach().float().cpu().numpy() if x.is_floating_point() else x.detach().cpu().numpy()
    return np.asarray(x)


def export_voice_clone_prompt(cache: VoiceRefCache, ref_audio_24k: np.ndarray, item, *, append_silence: bool,
                              model_identity: str, ref_text: str = "") -> str:
    """Store one upstream prompt item (attributes ``ref_spk_embedding``, ``ref_code``, ``ref_text``; reference
    ``model.py:336-352``) under the key of its audio.  Run once where upstream ``qwen-tts`` is available."""
    ref_code = getattr(item, "ref_code", None)
    key, meta = cache_key(ref_audio_24k, append_silence=append_silence, model_identity=model_identity,
                          mode="icl" if ref_code is not None else "xvec")
    text = getattr(item, "ref_text", None) or ref_text
    cache.save(key, meta, item.ref_spk_embedding, ref_code, ref_text=text or "")
    return key
