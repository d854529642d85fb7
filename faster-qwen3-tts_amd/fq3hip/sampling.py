"""Sampler entry points with the reference's signatures (``faster_qwen3_tts/sampling.py:10-66``),
executed by the HIP sampler kernel (``csrc/sampler_wave.cuh`` / ``sampler.cuh``).

Differences a caller can observe: tensors must live on the GPU; ``torch.multinomial``'s internal draw
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
    """In place, like the reference (sampling.py:10-29): ids in ``token_history`` get x/p (x>0) or x*p."""
    if repetition_penalty == 1.0 or token_history.numel() == 0:
        return logits
    if not logits.is_cuda:
        raise ValueError("fq3hip.sampling works on GPU tensors")
    ids = token_history.unique()
    t = logits[..., ids]
    logits[..., ids] = torch.where(t > 0, t / repetition_penalty, t * repetition_penalty)
    return logits


def sample_logits(logits: torch.Tensor, *, temperature: float, top_k: int, top_p: float, do_sample: bool,
                  suppress_mask: Optional[torch.Tensor] = None, suppress_tokens: Optional[Iterable[int]] = None,
                  noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """suppress -> temperature -> top-k (ties kept) -> top-p -> sample; returns LongTensor[batch]."""
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
    return torch.cat(out)
