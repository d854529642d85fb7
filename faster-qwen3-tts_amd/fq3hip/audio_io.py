"""WAV / PCM helpers shared by the CLI and the HTTP server (callers either side of the hot path).

The reference writes files with ``soundfile`` (``faster_qwen3_tts/cli.py:48-50``) and streams 16-bit PCM behind a WAV header
of unknown length (``examples/openai_server.py:93-119``).  ``soundfile`` is not part of this image, so files go through the
standard library's ``wave`` module (16-bit PCM, what ``sf.write`` produces for float input by default)."""
from __future__ import annotations

import io
import os
import struct
import wave
from typing import Tuple

import numpy as np


def to_pcm16(pcm: np.ndarray) -> bytes:
    """float waveform in [-1, 1] -> raw 16-bit little-endian PCM (examples/openai_server.py:93-95)."""
    return np.clip(np.asarray(pcm, dtype=np.float32) * 32768, -32768, 32767).astype("<i2").tobytes()


def wav_header(sample_rate: int, data_len: int = 0xFFFFFFFF) -> bytes:
    """RIFF/WAVE header for mono 16-bit PCM; ``data_len = 0xFFFFFFFF`` marks a stream of unknown size
    (examples/openai_server.py:98-113)."""
    n_channels, bits = 1, 16
    byte_rate = sample_rate * n_channels * bits // 8
    riff = 0xFFFFFFFF if data_len == 0xFFFFFFFF else 36 + data_len
    return (b"RIFF" + struct.pack("<I", riff) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, n_channels, sample_rate, byte_rate, n_channels * bits // 8, bits) +
            b"data" + struct.pack("<I", data_len))


def to_wav_bytes(pcm: np.ndarray, sample_rate: int) -> bytes:
    raw = to_pcm16(pcm)
    return wav_header(sample_rate, len(raw)) + raw


def write_wav(path: str, pcm: np.ndarray, sample_rate: int) -> None:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sample_rate))
        w.writeframes(to_pcm16(pcm))


def load_audio(path) -> Tuple[np.ndarray, int]:
    """Reference-clip loader shared by every caller (model.py:278-293 uses ``soundfile``): ``soundfile`` when it is
    installed (FLAC, IEEE-float WAV, ...), otherwise the standard-library PCM WAV reader with a clear error for what it cannot
    read.  -> (mono float32, sample_rate)."""
    try:
        import soundfile as sf
    except ImportError:
        return read_wav(str(path))
    audio, sr = sf.read(str(path), dtype="float32", always_2d=False)
    if audio.ndim > 1:
        audio = audio.mean(axis=1)
    return np.asarray(audio, dtype=np.float32), int(sr)


_PCM_SUBFORMAT = bytes.fromhex("0100000000001000800000aa00389b71")      # KSDATAFORMAT_SUBTYPE_PCM
_FLOAT_SUBFORMAT = bytes.fromhex("0300000000001000800000aa00389b71")    # KSDATAFORMAT_SUBTYPE_IEEE_FLOAT


def _wav_chunks(path: str):
    """(fmt fields, raw data bytes) of a RIFF/WAVE file, or (None, None) if it is not one.  fmt = dict(tag, channels, rate,
    block_align, bits, subformat) -- ``tag`` is the effective format: for WAVE_FORMAT_EXTENSIBLE (0xFFFE) the first two bytes of the
    SubFormat GUID when it is one of the two KSDATAFORMAT subtypes, else 0xFFFE."""
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            return None, None
        fmt, data = None, None
        while True:
            ck = f.read(8)
            if len(ck) < 8:
                break
            cid, size = ck[:4], struct.unpack("<I", ck[4:])[0]
            if cid == b"fmt ":
                body = f.read(size)
                if len(body) < 16:
                    return None, None
                tag, ch, rate, _bps, align, bits = struct.unpack("<HHIIHH", body[:16])
                sub = body[24:40] if tag == 0xFFFE and len(body) >= 40 else b""
                if tag == 0xFFFE and sub in (_PCM_SUBFORMAT, _FLOAT_SUBFORMAT):
                    tag = struct.unpack("<H", sub[:2])[0]
                fmt = dict(tag=tag, channels=ch, rate=rate, block_align=align, bits=bits)
                if size & 1:
                    f.seek(1, 1)
            elif cid == b"data":
                data = f.read(size)
                if size & 1:
                    f.seek(1, 1)
            else:
                f.seek(size + (size & 1), 1)
            if fmt is not None and data is not None:
                break
        return fmt, data


def _wav_format_tag(path: str) -> int:
    """Effective format tag of a RIFF/WAVE file (1 = integer PCM, 3 = IEEE float, 85 = MP3; an extensible header reports its
    SubFormat's tag), -1 if not RIFF/WAVE."""
    fmt, _ = _wav_chunks(path)
    return -1 if fmt is None else fmt["tag"]


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """Integer-PCM WAV (8 / 16 / 24 / 32 bits; plain or WAVE_FORMAT_EXTENSIBLE headers -- what ffmpeg and soundfile write for 24- and
    32-bit clips, and which the standard library's ``wave`` module refuses) -> (mono float32 in [-1, 1], sample_rate).  The chunks
    are parsed here; anything that is not integer PCM fails with a message that says what to do."""
    fmt, raw = _wav_chunks(str(path))
    tag = -1 if fmt is None else fmt["tag"]
    if tag != 1 or raw is None:
        what = {-1: "not a RIFF/WAVE file", 3: "IEEE-float WAV (format tag 3)", 85: "MP3-in-WAV (format tag 85)",
                0xFFFE: "WAVE_FORMAT_EXTENSIBLE with an unknown SubFormat"}.get(tag, f"WAV format tag {tag}")
        if tag == 1:
            what = "WAV file without a data chunk"
        raise ValueError(f"{path}: {what} cannot be read without the 'soundfile' package (not in this image); "
                         "convert the clip to 16-bit PCM WAV, or pass (waveform, sample_rate) / a voice_clone_prompt")
    ch, sr, bits = max(1, fmt["channels"]), fmt["rate"], fmt["bits"]
    sw = fmt["block_align"] // ch if fmt["block_align"] else (bits + 7) // 8      # container bytes per sample
    raw = raw[: len(raw) - len(raw) % (sw * ch)]
    if sw == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif sw == 3:
        b3 = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b3[:, 0] | (b3[:, 1] << 8) | (b3[:, 2] << 16)
        a = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif sw == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported WAV sample width {sw}")
    if ch > 1:
        a = a.reshape(-1, ch).mean(axis=1)
    return a, sr


def resample(audio: np.ndarray, sr: int, target_sr: int) -> np.ndarray:
    """Polyphase resampling on the host (``scipy.signal.resample_poly``).  Upstream resamples reference clips with librosa
    (soxr); the filters differ, so clips that are not already at ``target_sr`` give slightly different analyser inputs."""
    if int(sr) == int(target_sr):
        return np.asarray(audio, dtype=np.float32)
    from math import gcd
    from scipy.signal import resample_poly
    g = gcd(int(sr), int(target_sr))
    return resample_poly(np.asarray(audio, dtype=np.float64), int(target_sr) // g, int(sr) // g).astype(np.float32)
