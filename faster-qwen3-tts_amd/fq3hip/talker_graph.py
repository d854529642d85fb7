"""TalkerGraph with the reference's interface (``faster_qwen3_tts/talker_graph.py``) over the HIP context.

What the reference does with ``transformers.StaticCache`` + a captured ATen graph of ~500 kernels is
here a static KV cache owned by ``libfq3hip`` and 141 fused launches per token.  The per-position
additive mask table (talker_graph.py:71-92, a 2048-iteration Python loop) becomes two integers
(left-pad count, rope delta); attention reads only the live keys.
"""
from __future__ import annotations

from typing import Optional

import torch

from .engine import Fq3Engine


class TalkerGraph:
    def __init__(self, engine: Fq3Engine):
        self.engine = engine
        self.device = engine.device
        self.dtype = engine.dtype
        self.max_seq_len = engine.max_seq_len
        self.hidden_size = engine.cfg.talker.hidden_size
        self.num_layers = engine.cfg.talker.num_hidden_layers
        self.input_buf = torch.zeros(1, 1, self.hidden_size, dtype=self.dtype, device=self.device)
        self.output_buf = torch.zeros(1, 1, self.hidden_size, dtype=self.dtype, device=self.device)
        self.captured = False
        self._prefill_len = 0

    # talker_graph.py:109-147.  The decode-loop graph covers predictor + talker + sampler; capture once.
    @torch.inference_mode()
    def capture(self, prefill_len: int = 100, num_warmup: int = 3):
        self.engine.graph_capture()
        self.captured = True

    def reset(self, prefill_len: int = 0):
        self._prefill_len = 0

    def prefill_kv(self, past_key_values) -> int:
        """talker_graph.py:153-170.  Accepts an HF-style cache (indexable, layer -> (k, v) with
        [1, kv_heads, L, head_dim]) and copies it into the static slots, or an int when the prompt was
        prefilled by this context itself (``Fq3Engine.prefill`` writes the cache directly)."""
        if isinstance(past_key_values, int):
            if past_key_values > self.max_seq_len:
                raise RuntimeError(
                    f"Input is too long: prefill has {past_key_values} tokens but max_seq_len={self.max_seq_len}. "
                    "Use shorter text or shorter reference audio.")
            self._prefill_len = past_key_values
            return past_key_values
        seq_len = 0
        for li in range(self.num_layers):
            k, v = past_key_values[li]
            seq_len = k.shape[2]
            if seq_len > self.max_seq_len:
                raise RuntimeError(
                    f"Input is too long: prefill has {seq_len} tokens but max_seq_len={self.max_seq_len}. "
                    "Use shorter text or shorter reference audio.")
            self.engine.kv_import(li, k[0], v[0])
        self._prefill_len = seq_len
        return seq_len

    def set_generation_state(self, attention_mask: Optional[torch.Tensor], rope_deltas: Optional[torch.Tensor],
                             n_pad: Optional[int] = None):
        """talker_graph.py:172-196: pad-aware masking + rope delta.  ``n_pad``: the mask's pad count when the caller already has it
        (counting it from a device mask makes the host wait for the current stream)."""
        if n_pad is None:
            n_pad = 0
            if attention_mask is not None:
                n_pad = int((attention_mask[0] == 0).sum())
        delta = 0
        if rope_deltas is not None:
            delta = int(round(float(torch.as_tensor(rope_deltas).reshape(-1)[0])))
        elif n_pad:
            delta = -n_pad
        self.engine.set_generation_state(n_pad, delta)

    @torch.inference_mode()
    def run(self, input_embeds: torch.Tensor, position: int) -> torch.Tensor:
        """talker_graph.py:198-214: returns the static output buffer [1, 1, H] (clone to keep)."""
        self.input_buf.copy_(input_embeds.reshape(1, 1, -1))
        self.engine.talker_step(self.input_buf.view(-1), int(position), out=self.output_buf.view(-1))
        return self.output_buf
