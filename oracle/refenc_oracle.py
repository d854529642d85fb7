"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference-audio analysers (never imported by the product).

What it restates: the two third-party modules upstream's ``create_voice_clone_prompt`` runs for a new reference clip (the
reference wrapper's call site is ``faster_qwen3_tts/model.py:430-447``; ``qwen-tts`` itself is absent offline):

* ``speech_tokenizer.encode`` -- the 12 Hz tokenizer's encoder, a ``MimiModel`` encoder [recalled: upstream's
  ``Qwen3TTSTokenizerV2Encoder`` subclasses it and keeps the first 16 quantizers].  Restated from
  ``transformers/models/mimi/modeling_mimi.py`` (5.x as installed): conv padding ``:269-345``, SEANet encoder ``:450-492``,
  transformer layer ``:729-779`` (attention ``:657-726``, MLP ``:602-616``, layer scale ``:495-508``), frame encode
  ``:1231-1268``, Euclidean codebook / RVQ / split RVQ ``:964-1138``.
* ``extract_speaker_embedding`` -- BigVGAN-style log-mel (reflect pad (n_fft-hop)/2, Hann STFT, sqrt(|X|^2 + 1e-9),
  Slaney mel basis, log(clamp 1e-5)) [recalled] followed by ECAPA-TDNN, restated from
  ``transformers/models/qwen2_5_omni/modeling_qwen2_5_omni.py:2412-2707``.

PINNED by ``tests/test_refenc_oracle_pins.py`` against those transformers modules instantiated with the same seeded
weights (both are importable in the build container and on the GPU box); the mapping of Qwen3-TTS checkpoints onto these
architectures stays [recalled] until a real checkpoint is available.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

Weights = Dict[str, torch.Tensor]


# ---- speech-tokenizer encoder ----------------------------------------------------------------------------------------------
def mimi_conv1d(x, w, b, stride=1, dilation=1, pad_mode="constant"):
    """Causal MimiConv1d.forward (modeling_mimi.py:269-277, :327-345): left pad = effective kernel - stride, right pad up
    to a whole number of frames."""
    k_eff = (w.shape[-1] - 1) * dilation + 1
    pt = k_eff - stride
    L = x.shape[-1]
    n_frames = math.ceil((L - k_eff + pt) / stride + 1) - 1
    extra = n_frames * stride + k_eff - pt - L
    x = F.pad(x, (pt, extra), mode=pad_mode)
    return F.conv1d(x, w, b, stride=stride, dilation=dilation)


def seanet_encoder(W: Weights, rc, x):
    """MimiEncoder.forward (modeling_mimi.py:450-492) with MimiResnetBlock (:408-447). x [1, 1, n]."""
    E = "encoder.encoder.layers"
    g = lambda n: W[n].to(x.dtype)
    x = mimi_conv1d(x, g(f"{E}.0.conv.weight"), g(f"{E}.0.conv.bias"))
    li = 1
    for r in rc.ratios:
        for j in range(rc.num_residual_layers):
            h = F.elu(x)
            h = mimi_conv1d(h, g(f"{E}.{li}.block.1.conv.weight"), g(f"{E}.{li}.block.1.conv.bias"), dilation=rc.dilation_growth_rate ** j)
            h = F.elu(h)
            h = mimi_conv1d(h, g(f"{E}.{li}.block.3.conv.weight"), g(f"{E}.{li}.block.3.conv.bias"))
            x = x + h
            li += 1
        li += 1
        x = mimi_conv1d(F.elu(x), g(f"{E}.{li}.conv.weight"), g(f"{E}.{li}.conv.bias"), stride=r)
        li += 1
    li += 1
    return mimi_conv1d(F.elu(x), g(f"{E}.{li}.conv.weight"), g(f"{E}.{li}.conv.bias"))


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def encoder_transformer(W: Weights, rc, x):
    """MimiTransformerModel over [T, hidden] (modeling_mimi.py:729-779, :657-726): pre-LayerNorm, RoPE, causal
    sliding-window attention (key j visible to query i iff 0 <= i - j < window), GELU MLP, layer scale."""
    T = x.shape[0]
    nh, hd = rc.num_attention_heads, rc.head_dim
    g = lambda n: W[n].to(x.dtype)
    inv = 1.0 / (rc.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    ang = torch.arange(T, dtype=torch.float32)[:, None] * inv[None, :]
    emb = torch.cat((ang, ang), dim=-1)
    cos, sin = emb.cos().to(x.dtype), emb.sin().to(x.dtype)
    i = torch.arange(T)
    visible = (i[None, :] <= i[:, None]) & (i[:, None] - i[None, :] < rc.sliding_window)
    for l in range(rc.num_hidden_layers):
        p = f"encoder.encoder_transformer.layers.{l}"
        h = F.layer_norm(x, (x.shape[-1],), g(f"{p}.input_layernorm.weight"), g(f"{p}.input_layernorm.bias"), rc.norm_eps)
        q = F.linear(h, g(f"{p}.self_attn.q_proj.weight")).view(T, nh, hd).transpose(0, 1)
        k = F.linear(h, g(f"{p}.self_attn.k_proj.weight")).view(T, nh, hd).transpose(0, 1)
        v = F.linear(h, g(f"{p}.self_attn.v_proj.weight")).view(T, nh, hd).transpose(0, 1)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        s = s.masked_fill(~visible, float("-inf"))
        a = torch.softmax(s.float(), dim=-1).to(x.dtype) @ v
        a = a.transpose(0, 1).reshape(T, nh * hd)
        x = x + g(f"{p}.self_attn_layer_scale.scale") * F.linear(a, g(f"{p}.self_attn.o_proj.weight"))
        h = F.layer_norm(x, (x.shape[-1],), g(f"{p}.post_attention_layernorm.weight"), g(f"{p}.post_attention_layernorm.bias"), rc.norm_eps)
        h = F.linear(F.gelu(F.linear(h, g(f"{p}.mlp.fc1.weight"))), g(f"{p}.mlp.fc2.weight"))
        x = x + g(f"{p}.mlp_layer_scale.scale") * h
    return x


def rvq_encode(W: Weights, rc, emb, dtype=torch.float32):
    """MimiSplitResidualVectorQuantizer.encode (modeling_mimi.py:1097-1124) over emb [hidden, T]; per level the nearest
    codebook row (:984-989) and ``residual -= row`` (:1072-1077).  Returns codes [T, nq] and, per decision, the relative
    gap between the two smallest squared distances (how close the arg-min was)."""
    T = emb.shape[-1]
    codes = torch.zeros(T, rc.num_quantizers, dtype=torch.long)
    margins = torch.zeros(T, rc.num_quantizers, dtype=torch.float64)
    lv = 0
    for name, n in (("semantic", rc.num_semantic_quantizers), ("acoustic", rc.num_quantizers - rc.num_semantic_quantizers)):
        q = f"encoder.quantizer.{name}_residual_vector_quantizer"
        res = F.conv1d(emb[None], W[f"{q}.input_proj.weight"].to(dtype))[0].transpose(0, 1)          # [T, D]
        for i in range(n):
            usage = W[f"{q}.layers.{i}.codebook.cluster_usage"].float().clamp(min=1e-5)
            book = (W[f"{q}.layers.{i}.codebook.embed_sum"].float() / usage[:, None]).to(dtype)      # :980-983
            d2 = ((res[:, None, :].double() - book[None].double()) ** 2).sum(-1)                     # exact ordering of cdist
            two = torch.topk(d2, 2, dim=-1, largest=False)
            idx = d2.argmin(-1)
            codes[:, lv] = idx
            margins[:, lv] = (two.values[:, 1] - two.values[:, 0]) / two.values[:, 1].clamp(min=1e-30)
            res = res - book[idx]
            lv += 1
    return codes, margins


def tokenizer_encode(W: Weights, rc, wav: torch.Tensor, dtype=torch.float32, return_all: bool = False):
    """MimiModel._encode_frame (modeling_mimi.py:1231-1268): waveform [n] -> codes [T, nq]."""
    x = wav.reshape(1, 1, -1).to(dtype)
    h = seanet_encoder(W, rc, x)                                   # [1, hidden, T25]
    h = encoder_transformer(W, rc, h[0].transpose(0, 1))           # [T25, hidden]
    d = mimi_conv1d(h.transpose(0, 1)[None], W["encoder.downsample.conv.weight"].to(dtype), None, stride=2, pad_mode="replicate")[0]
    codes, margins = rvq_encode(W, rc, d, dtype)
    return (codes, margins, h, d) if return_all else codes


def encoded_length(rc, n: int) -> int:
    """MimiModel.get_encoded_length (:1270-1284): every strided causal conv produces ceil(L / stride) frames."""
    for r in rc.ratios:
        n = -(-n // r)
    return -(-n // 2)


# ---- speaker encoder ------------------------------------------------------------------------------------------------------------
def slaney_mel(sr, n_fft, n_mels, fmin, fmax):
    """``librosa.filters.mel`` with its defaults (Slaney scale, area normalisation), written out."""
    def h2m(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), f * 3.0 / 200.0)

    def m2h(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * 200.0 / 3.0)

    freqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    pts = m2h(np.linspace(h2m(fmin), h2m(fmax), n_mels + 2))
    out = np.zeros((n_mels, len(freqs)))
    for i in range(n_mels):
        lo, ce, hi = pts[i], pts[i + 1], pts[i + 2]
        up = (freqs - lo) / (ce - lo)
        down = (hi - freqs) / (hi - ce)
        out[i] = np.maximum(0, np.minimum(up, down)) * 2.0 / (hi - lo)
    return torch.from_numpy(out)


def mel_spectrogram(rc, wav: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """BigVGAN-style log-mel [frames, n_mels]."""
    y = wav.reshape(1, -1).to(dtype)
    pad = (rc.n_fft - rc.hop_size) // 2
    y = F.pad(y[None], (pad, pad), mode="reflect")[0]
    spec = torch.stft(y, rc.n_fft, hop_length=rc.hop_size, win_length=rc.n_fft, window=torch.hann_window(rc.n_fft, dtype=dtype),
                      center=False, normalized=False, onesided=True, return_complex=True)
    mag = torch.sqrt(torch.view_as_real(spec).pow(2).sum(-1) + 1e-9)[0]            # [bins, frames]
    mel = slaney_mel(rc.sample_rate, rc.n_fft, rc.mel_dim, rc.fmin, rc.fmax).to(dtype) @ mag
    return torch.log(torch.clamp(mel, min=1e-5)).transpose(0, 1)


def _tdnn(W, name, x, k, dil):
    """TimeDelayNetBlock (modeling_qwen2_5_omni.py:2412-2432): Conv1d(padding="same", padding_mode="reflect") + ReLU."""
    w, b = W[name + ".weight"].to(x.dtype), W[name + ".bias"].to(x.dtype)
    pad = (k - 1) * dil // 2
    if pad:
        x = F.pad(x, (pad, pad), mode="reflect")
    return F.relu(F.conv1d(x, w, b, dilation=dil))


def ecapa(W: Weights, rc, mel: torch.Tensor):
    """ECAPA_TimeDelayNet.forward (modeling_qwen2_5_omni.py:2630-2707) over mel [frames, n_mels] -> [enc_dim]."""
    S = "speaker_encoder"
    x = mel.transpose(0, 1)[None]                                                   # [1, mel, T]
    ks, ds = rc.enc_kernel_sizes, rc.enc_dilations
    x = _tdnn(W, f"{S}.blocks.0.conv", x, ks[0], ds[0])
    outs = []
    for b in range(1, len(rc.enc_channels) - 1):
        B = f"{S}.blocks.{b}"
        res = x
        h = _tdnn(W, f"{B}.tdnn1.conv", x, 1, 1)
        parts, prev = [], None                                                       # Res2NetBlock :2435-2466
        for i, part in enumerate(torch.chunk(h, rc.enc_res2net_scale, dim=1)):
            if i == 0:
                prev = part
            elif i == 1:
                prev = _tdnn(W, f"{B}.res2net_block.blocks.{i - 1}.conv", part, ks[b], ds[b])
            else:
                prev = _tdnn(W, f"{B}.res2net_block.blocks.{i - 1}.conv", part + prev, ks[b], ds[b])
            parts.append(prev)
        h = _tdnn(W, f"{B}.tdnn2.conv", torch.cat(parts, dim=1), 1, 1)
        m = h.mean(dim=2, keepdim=True)                                              # SqueezeExcitationBlock :2469-2496
        m = F.relu(F.conv1d(m, W[f"{B}.se_block.conv1.weight"].to(x.dtype), W[f"{B}.se_block.conv1.bias"].to(x.dtype)))
        m = torch.sigmoid(F.conv1d(m, W[f"{B}.se_block.conv2.weight"].to(x.dtype), W[f"{B}.se_block.conv2.bias"].to(x.dtype)))
        x = h * m + res
        outs.append(x)
    h = _tdnn(W, f"{S}.mfa.conv", torch.cat(outs, dim=1), ks[-1], ds[-1])
    T = h.shape[-1]                                                                  # AttentiveStatisticsPooling :2499-2585
    mean = h.mean(dim=2)
    std = torch.sqrt(((h - mean[..., None]) ** 2).mean(dim=2).clamp(1e-12))
    att = torch.cat([h, mean[..., None].expand(-1, -1, T), std[..., None].expand(-1, -1, T)], dim=1)
    att = torch.tanh(_tdnn(W, f"{S}.asp.tdnn.conv", att, 1, 1))
    att = F.conv1d(att, W[f"{S}.asp.conv.weight"].to(x.dtype), W[f"{S}.asp.conv.bias"].to(x.dtype))
    att = torch.softmax(att, dim=2)
    mean = (att * h).sum(2)
    std = torch.sqrt((att * (h - mean[..., None]) ** 2).sum(2).clamp(1e-12))
    pooled = torch.cat([mean, std], dim=1)[..., None]
    return F.conv1d(pooled, W[f"{S}.fc.weight"].to(x.dtype), W[f"{S}.fc.bias"].to(x.dtype))[0, :, 0]


def speaker_embedding(W: Weights, rc, wav: torch.Tensor, dtype=torch.float32):
    mel = mel_spectrogram(rc, wav, dtype)
    return ecapa(W, rc, mel), mel
