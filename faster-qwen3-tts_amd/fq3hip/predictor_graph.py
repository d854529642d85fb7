"""PredictorGraph with the reference's interface (``faster_qwen3_tts/predictor_graph.py``): the
15-codebook loop (2-token prefill, 14 single-token passes, 15 heads + samplers) as fused HIP launches
with the sampler on the device -- no host round trip inside the loop."""
from __future__ import annotations

from typing import Optional

import torch

from .engine import Fq3Engine


class PredictorGraph:
    def __init__(self, engine: Fq3Engine, do_sample: bool = True, top_k: int = 50, top_p: float = 1.0,
                 temperature: float = 0.9):
        self.engine = engine
        self.device, self.dtype = engine.device, engine.dtype
        self.num_layers = engine.cfg.predictor.num_hidden_layers
        self.hidden_size = engine.cfg.predictor.hidden_size
        self.num_code_groups = engine.cfg.num_code_groups
        self.num_codebooks = self.num_code_groups - 1
        self.max_seq = 2 + self.num_codebooks
        self._policy = dict(do_sample=bool(do_sample), top_k=int(top_k), top_p=float(top_p), temperature=float(temperature))
        self.captured = False
        self._push()

    def _push(self):
        self.engine.set_predictor_sampling(**self._policy)

    # policy attributes are settable before capture, like the reference's plain attributes
    # (tests/test_e2e_parity.py:208-215 forces greedy this way)
    def _get(name):
        return property(lambda self: self._policy[name],
                        lambda self, v: (self._policy.__setitem__(name, type(self._policy[name])(v)), self._push())[1])
    do_sample = _get("do_sample")
    top_k = _get("top_k")
    top_p = _get("top_p")
    temperature = _get("temperature")
    del _get

    @torch.inference_mode()
    def capture(self, num_warmup: int = 3):
        self.captured = True       # the loop graph is captured by TalkerGraph.capture (one graph per frame)

    @torch.inference_mode()
    def run(self, pred_input: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """pred_input [1, 2, H_talker] -> fresh LongTensor[15] (predictor_graph.py:204-214)."""
        x = pred_input.reshape(-1).to(self.dtype).contiguous()
        if self._policy["do_sample"] and noise is None:
            noise = torch.empty(self.num_codebooks, self.engine.cfg.predictor.vocab_size, dtype=self.dtype,
                                device=self.device).exponential_(1)
        return self.engine.predictor_loop(x, noise=noise)
