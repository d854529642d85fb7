"""HIP reference-audio analysers: ``speech_tokenizer.encode`` and ``extract_speaker_embedding``.

The reference computes a voice-clone prompt for a new ``(ref_audio, ref_text)`` pair through upstream's
``create_voice_clone_prompt`` (``faster_qwen3_tts/model.py:430-447``) and caches it (``:424-463``); upstream runs the
speech tokenizer's encoder (reference codes for ICL mode) and the speaker encoder (x-vector) there.  This module is the
host side of ``csrc/fq3_refenc.hip``: it re-lays the checkpoint tensors out once (GEMM-ready conv weights, the windowed-DFT
and mel tables) and exposes the two calls.  No CPU path: without ``libfq3hip.so`` construction raises ``ImportError``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib as L
from .config import RefAudioConfig

Weights = Dict[str, torch.Tensor]


def slaney_mel_basis(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """Slaney-style (area-normalised, linear below 1 kHz / log above) mel filterbank, ``[n_mels, n_fft // 2 + 1]``:
    the ``librosa.filters.mel`` defaults BigVGAN-style ``mel_spectrogram`` front ends use."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    wts = np.maximum(0.0, np.minimum(lower, upper))
    wts *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return wts.astype(np.float32)


def dft_table(n_fft: int, n_bins_padded: int) -> torch.Tensor:
    """``[2 * NB, n_fft]``: rows ``k < n_fft/2+1`` = periodic-Hann * cos(2 pi k n / N), rows ``NB + k`` = Hann * sin;
    the padding rows are zero (computed in float64, stored fp32)."""
    n = np.arange(n_fft, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)                     # torch.hann_window(periodic=True)
    k = np.arange(n_fft // 2 + 1, dtype=np.float64)[:, None]
    ang = 2.0 * np.pi * k * n[None, :] / n_fft
    tab = np.zeros((2 * n_bins_padded, n_fft), dtype=np.float64)
    tab[: n_fft // 2 + 1] = win * np.cos(ang)
    tab[n_bins_padded: n_bins_padded + n_fft // 2 + 1] = win * np.sin(ang)
    return torch.from_numpy(tab.astype(np.float32))


def pack_ref_audio_weights(W: Weights, rc: RefAudioConfig) -> Weights:
    """checkpoint layout -> the layouts documented in ``csrc/fq3_refenc.hip`` (host glue, runs once; fp32)."""
    out: Weights = {}
    f = lambda t: t.detach().to(torch.float32)
    conv = lambda w: f(w).permute(0, 2, 1).contiguous()                    # [Cout, Cin, k] -> [Cout, k, Cin]

    def strided(w, r):                                                     # [Cout, Cin, 2r] -> [Cout, 2, r*Cin]
        co, ci, k = w.shape
        return f(w).reshape(co, ci, 2, r).permute(0, 2, 3, 1).reshape(co, 2, r * ci).contiguous()

    if any(k.startswith("encoder.") for k in W):
        E = "encoder.encoder.layers"
        out[f"{E}.0.conv.weight"] = f(W[f"{E}.0.conv.weight"]).reshape(rc.num_filters, rc.kernel_size).contiguous()
        out[f"{E}.0.conv.bias"] = f(W[f"{E}.0.conv.bias"])
        li = 1
        for r in rc.ratios:
            for _ in range(rc.num_residual_layers):
                for j in (1, 3):
                    out[f"{E}.{li}.block.{j}.conv.weight"] = conv(W[f"{E}.{li}.block.{j}.conv.weight"])
                    out[f"{E}.{li}.block.{j}.conv.bias"] = f(W[f"{E}.{li}.block.{j}.conv.bias"])
                li += 1
            li += 1
            out[f"{E}.{li}.conv.weight"] = strided(W[f"{E}.{li}.conv.weight"], r)
            out[f"{E}.{li}.conv.bias"] = f(W[f"{E}.{li}.conv.bias"])
            li += 1
        li += 1
        out[f"{E}.{li}.conv.weight"] = conv(W[f"{E}.{li}.conv.weight"])
        out[f"{E}.{li}.conv.bias"] = f(W[f"{E}.{li}.conv.bias"])
        for l in range(rc.num_hidden_layers):
            p = f"encoder.encoder_transformer.layers.{l}"
            out[f"{p}.self_attn.qkv.weight"] = torch.cat([f(W[f"{p}.self_attn.{n}.weight"]) for n in ("q_proj", "k_proj", "v_proj")], 0).contiguous()
            for n in ("self_attn.o_proj.weight", "mlp.fc1.weight", "mlp.fc2.weight", "input_layernorm.weight", "input_layernorm.bias",
                      "post_attention_layernorm.weight", "post_attention_layernorm.bias", "self_attn_layer_scale.scale", "mlp_layer_scale.scale"):
                out[f"{p}.{n}"] = f(W[f"{p}.{n}"]).contiguous()
        out["encoder.downsample.conv.weight"] = strided(W["encoder.downsample.conv.weight"], 2)
        q = "encoder.quantizer.{}_residual_vector_quantizer"
        out["encoder.quantizer.input_proj.weight"] = torch.cat([f(W[q.format(n) + ".input_proj.weight"]).squeeze(-1) for n in ("semantic", "acoustic")], 0).contiguous()
        for lv in range(rc.num_quantizers):
            sem = lv < rc.num_semantic_quantizers
            src = q.format("semantic" if sem else "acoustic") + f".layers.{lv if sem else lv - rc.num_semantic_quantizers}.codebook."
            out[f"encoder.quantizer.codebook.{lv}.embed_sum"] = f(W[src + "embed_sum"]).contiguous()
            out[f"encoder.quantizer.codebook.{lv}.cluster_usage"] = f(W[src + "cluster_usage"]).contiguous()
        half = rc.head_dim // 2
        inv = 1.0 / (rc.rope_theta ** (torch.arange(0, half, dtype=torch.float32) / half))
        ang = torch.arange(rc.max_positions, dtype=torch.float32)[:, None] * inv[None, :]
        out["encoder.rope.cos"], out["encoder.rope.sin"] = ang.cos().contiguous(), ang.sin().contiguous()
    if any(k.startswith("speaker_encoder.") for k in W):
        S = "speaker_encoder"
        NB = rc.n_bins_padded
        out[f"{S}.mel.dft"] = dft_table(rc.n_fft, NB)
        mb = torch.zeros(rc.mel_dim, NB)
        mb[:, : rc.n_fft // 2 + 1] = torch.from_numpy(slaney_mel_basis(rc.sample_rate, rc.n_fft, rc.mel_dim, rc.fmin, rc.fmax))
        out[f"{S}.mel.basis"] = mb
        n_enc = len(rc.enc_channels)
        names = [f"{S}.blocks.0.conv", f"{S}.mfa.conv", f"{S}.asp.conv", f"{S}.fc"]
        for b in range(1, n_enc - 1):
            B = f"{S}.blocks.{b}"
            names += [f"{B}.tdnn1.conv", f"{B}.tdnn2.conv", f"{B}.se_block.conv1", f"{B}.se_block.conv2"]
            names += [f"{B}.res2net_block.blocks.{i}.conv" for i in range(rc.enc_res2net_scale - 1)]
        for n in names:
            out[n + ".weight"] = conv(W[n + ".weight"])
            out[n + ".bias"] = f(W[n + ".bias"])
        cm = rc.enc_channels[-1]
        wa = f(W[f"{S}.asp.tdnn.conv.weight"]).squeeze(-1)                 # [A, 3*Cm]: columns = [h | mean | std]
        out[f"{S}.asp.tdnn.conv.weight_h"] = wa[:, :cm].contiguous()
        out[f"{S}.asp.tdnn.conv.weight_ms"] = wa[:, cm:].contiguous()
        out[f"{S}.asp.tdnn.conv.bias"] = f(W[f"{S}.asp.tdnn.conv.bias"])
    return out


class HipRefAudioAnalyzer:
    """24 kHz mono waveform -> reference codes / speaker embedding, on the HIP kernels."""

    def __init__(self, rc: RefAudioConfig, weights: Weights, device: str = "cuda"):
        self.lib = L.load()
        self.cfg, self.device = rc, torch.device(device)
        self.sample_rate = int(rc.sample_rate)
        cc = L.RefEncConfig()
        cc.num_filters, cc.n_ratios = rc.num_filters, len(rc.ratios)
        for i, v in enumerate(rc.ratios):
            cc.ratios[i] = v
        cc.kernel_size, cc.last_kernel_size, cc.residual_kernel_size = rc.kernel_size, rc.last_kernel_size, rc.residual_kernel_size
        cc.n_residual_layers, cc.dilation_growth_rate, cc.compress = rc.num_residual_layers, rc.dilation_growth_rate, rc.compress
        cc.hidden, cc.n_layers, cc.n_heads, cc.head_dim = rc.hidden_size, rc.num_hidden_layers, rc.num_attention_heads, rc.head_dim
        cc.inter, cc.sliding_window, cc.norm_eps = rc.intermediate_size, rc.sliding_window, rc.norm_eps
        cc.num_quantizers, cc.num_semantic = rc.num_quantizers, rc.num_semantic_quantizers
        cc.codebook_size, cc.codebook_dim, cc.max_positions = rc.codebook_size, rc.codebook_dim, rc.max_positions
        cc.mel_dim, cc.n_fft, cc.hop, cc.n_bins_padded = rc.mel_dim, rc.n_fft, rc.hop_size, rc.n_bins_padded
        cc.n_enc = len(rc.enc_channels)
        for i in range(cc.n_enc):
            cc.enc_channels[i], cc.enc_kernel_sizes[i], cc.enc_dilations[i] = rc.enc_channels[i], rc.enc_kernel_sizes[i], rc.enc_dilations[i]
        cc.attn_channels, cc.res2net_scale, cc.se_channels, cc.enc_dim = rc.enc_attention_channels, rc.enc_res2net_scale, rc.enc_se_channels, rc.enc_dim
        self.h = L.vp()
        with torch.cuda.device(self.device):
            L.check(self.lib.fq3_refenc_create(C.byref(cc), C.byref(self.h)))
        self._bound = {k: v.to(device=self.device, dtype=torch.float32).contiguous() for k, v in pack_ref_audio_weights(weights, rc).items()}
        for name, t in self._bound.items():
            L.check(self.lib.fq3_refenc_bind(self.h, name.encode(), t.data_ptr(), t.numel()))
        self.has_encoder = any(k.startswith("encoder.") for k in self._bound)
        self.has_speaker = any(k.startswith("speaker_encoder.") for k in self._bound)
        with torch.cuda.device(self.device):
            L.check(self.lib.fq3_refenc_finalize(self.h, torch.cuda.current_stream(self.device).cuda_stream))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.fq3_refenc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _wave(self, wav) -> torch.Tensor:
        t = torch.as_tensor(np.asarray(wav, dtype=np.float32) if not isinstance(wav, torch.Tensor) else wav)
        return t.reshape(-1).to(device=self.device, dtype=torch.float32).contiguous()

    def num_frames(self, n_samples: int) -> int:
        return int(self.lib.fq3_refenc_num_frames(self.h, int(n_samples)))

    def encode(self, wav) -> torch.Tensor:
        """``speech_tokenizer.encode``: waveform -> ``LongTensor[T, num_quantizers]`` (12.5 frames per second)."""
        x = self._wave(wav)
        T = self.num_frames(x.numel())
        codes = torch.empty(T, self.cfg.num_quantizers, dtype=torch.long, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.fq3_refenc_encode(self.h, x.data_ptr(), x.numel(), codes.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream))
        return codes

    def speaker_embedding(self, wav, return_mel: bool = False):
        """``extract_speaker_embedding``: waveform -> ``FloatTensor[enc_dim]`` (and the log-mel frames it was computed from)."""
        x = self._wave(wav)
        emb = torch.empty(self.cfg.enc_dim, dtype=torch.float32, device=self.device)
        mel = None
        if return_mel:
            n_frames = (x.numel() + 2 * ((self.cfg.n_fft - self.cfg.hop_size) // 2) - self.cfg.n_fft) // self.cfg.hop_size + 1
            mel = torch.empty(n_frames, self.cfg.mel_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.fq3_refenc_speaker(self.h, x.data_ptr(), x.numel(), emb.data_ptr(), mel.data_ptr() if mel is not None else None,
                                                torch.cuda.current_stream(self.device).cuda_stream))
        return (emb, mel) if return_mel else emb
