"""Sampler entry points with the reference's signatures (``faster_qwen3_tts/sampling.py:10-66``),
executed by the HIP sampler kernels (``csrc/sampler_wave.cuh`` / ``sampler.cuh``, ``fq3_sample`` and
``fq3_apply_repetition_penalty``).  The decode loop never calls these Python functions (penalty, suppression and
sampling are fused into the frame graph); they exist for callers of the reference's module-level API.

Differences a caller can observe: ``sample_logits`` needs GPU tensors; ``torch.multinomial``'s internal draw
is replaced by an explicit Exp(1) ``noise`` tensor (drawn here with ``Tensor.exponential_`` when not
given, i.e. the very variates ``multinomial`` would draw); an arbitrary ``suppress_mask`` is applied
with one masked_fill before the kernel, the canonical "[V-1024, V) except EOS" range used by the
decode loop is applied inside it.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Tuple

import torch

_SAMPLERS: Dict[Tuple[int, torch.dtype], object] = {}


def _sampler(device: torch.device, dtype: torch.dtype):
    """A weight-less context used only for its sampler kernels."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), dtype)
    if key not in _SAMPLERS:
        from .engine import Fq3Engine
        _SAMPLERS[key] = Fq3Engine.sampler_only(device, dtype)
    return _SAMPLERS[key]


def apply_repetition_penalty(logits: torch.Tensor, token_history: torch.Tensor, repetition_penalty: float) -> torch.Tensor:
    """In place, like the reference (sampling.py:10-29): ids in ``token_history`` get x/p (x>0) or x*p.

    GPU rows of a supported dtype go through ``fq3_apply_repetition_penalty``; anything else (the CPU tensors of the
    reference's own unit test, tests/test_sampling.py:10-21) is plain tensor arithmetic on a history mask -- host-side
    API compatibility, not part of the decode path."""
    if repetition_penalty == 1.0 or token_history.numel() == 0:
        return logits
    V = logits.shape[-1]
    if logits.is_cuda and logits.dtype in (torch.bfloat16, torch.float32) and logits.is_contiguous() and V <= 4096:
        import ctypes as C
        from . import _lib as L
        eng = _sampler(logits.device, logits.dtype)
        hist = token_history.to(device=logits.device, dtype=torch.long).contiguous().view(-1)
        rows = logits.view(-1, V)
        for b in range(rows.shape[0]):
            L.check(eng.lib.fq3_apply_repetition_penalty(eng.ctx, rows[b].data_ptr(), int(V), hist.data_ptr(), int(hist.numel()),
                                                         C.c_float(float(repetition_penalty)), eng._stream()))
        return logits
    seen = torch.zeros(V, dtype=torch.bool, device=logits.device)
    seen[token_history.reshape(-1).to(logits.device)] = True
    scaled = torch.where(logits > 0, logits / repetition_penalty, logits * repetition_penalty)
    logits.copy_(torch.where(seen, scaled, logits))
    return logits


def sample_logits(logits: torch.Tensor, *, temperature: float, top_k: int, top_p: float, do_sample: bool,
                  suppress_mask: Optional[torch.Tensor] = None, suppress_tokens: Optional[Iterable[int]] = None,
                  noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """suppress -> temperature -> top-k (ties kept) -> top-p -> sample; returns LongTensor of shape ``logits.shape[:-1]``."""
    if not logits.is_cuda:
        raise ValueError("fq3hip.sampling works on GPU tensors (the HIP path has no CPU fallback)")
    if logits.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("logits must be bfloat16 or float32")
    x = logits.reshape(-1, logits.shape[-1])
    if suppress_mask is not None or suppress_tokens:
        x = x.clone()
        if suppress_mask is not None:
            x[..., suppress_mask] = float("-inf")
        if suppress_tokens:
            x[..., list(suppress_tokens)] = float("-inf")
    eng = _sampler(logits.device, logits.dtype)
    out = []
    for b in range(x.shape[0]):
        row = x[b].contiguous()
        nz = None
        if do_sample:
            nz = (noise.reshape(-1, row.numel())[b] if noise is not None else torch.empty_like(row).exponential_(1)).contiguous()
        out.append(eng.sample(row, temperature=temperature, top_k=top_k, top_p=top_p, do_sample=do_sample, noise=nz))
    return torch.cat(out).reshape(logits.shape[:-1])
