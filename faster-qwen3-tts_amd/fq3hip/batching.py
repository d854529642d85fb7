"""Continuous batching over ``Fq3Batch`` (``fq3_batch_*``): several utterances decode in lock-step over one weight
stream; a finished lane is re-armed with the next request at a frame boundary.

No reference equivalent -- the reference decodes one utterance at a time (``talker_graph.py:46`` /
``predictor_graph.py:70`` fix batch = 1 and ``examples/openai_server.py:71`` serialises requests with a lock); the
per-utterance semantics are the reference's (``generate.py:16-215``): same prefill, same first-token sampling, same
suppress / min_new / repetition-penalty policy, and -- because the batch kernels keep the single-stream arithmetic --
the same ids as ``fast_generate`` for the same noise.
"""
from __future__ import annotations

import time
from collections import deque
from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, Iterator, List, Optional, Tuple

import torch

from .generate import NOISE_RING, _prefill_and_arm, _refill
from .predictor_graph import PredictorGraph
from .talker_graph import TalkerGraph


@dataclass
class BatchRequest:
    """One utterance: the tensors ``fast_generate`` takes plus its sampling arguments."""
    rid: Any
    talker: Any
    talker_input_embeds: torch.Tensor
    attention_mask: torch.Tensor
    trailing_text_hiddens: torch.Tensor
    tts_pad_embed: torch.Tensor
    config: Any
    gen_kwargs: Dict[str, Any] = field(default_factory=dict)


@dataclass
class _Lane:
    index: int
    engine: Any
    talker_graph: Any
    predictor_graph: Any
    req: Optional[BatchRequest] = None
    tn: Optional[torch.Tensor] = None
    pn: Optional[torch.Tensor] = None
    issued: int = 0
    max_frames: int = 0
    t_arm: float = 0.0
    prefill_ms: float = 0.0


class BatchDecoder:
    """Drives up to ``len(engines)`` lanes.  ``engines[1:]`` must share ``engines[0]``'s weights
    (``Fq3Engine(..., share=engines[0])``)."""

    def __init__(self, engines: List[Any], predictor_policy: Optional[Dict[str, Any]] = None, poll_every: int = 8,
                 use_graph: bool = True, batch_factory=None):
        if batch_factory is None:
            from .engine import Fq3Batch as batch_factory
        policy = predictor_policy or dict(do_sample=True, top_k=50, top_p=1.0, temperature=0.9)
        self.lanes = [_Lane(i, e, TalkerGraph(e), PredictorGraph(e, **policy)) for i, e in enumerate(engines)]
        self.batch = batch_factory(engines)
        self.poll_every = max(1, int(poll_every))
        self.use_graph = use_graph
        self._captured = False

    def set_predictor_policy(self, **policy):
        for ln in self.lanes:
            for k, v in policy.items():
                setattr(ln.predictor_graph, k, v)

    # ------------------------------------------------------------------------------------------------------------
    def _arm(self, ln: _Lane, req: BatchRequest):
        kw = dict(max_new_tokens=2048, min_new_tokens=2, temperature=0.9, top_k=50, top_p=1.0, do_sample=True,
                  repetition_penalty=1.05)
        kw.update(req.gen_kwargs)
        if float(kw["top_p"]) < 1.0 or float(ln.predictor_graph.top_p) < 1.0:
            raise NotImplementedError("batched decode supports top_p >= 1.0 only; use the single-stream path for nucleus sampling")
        t0 = time.time()
        _eng, tn, pn, max_frames = _prefill_and_arm(
            req.talker, req.talker_input_embeds, req.attention_mask, req.trailing_text_hiddens, req.tts_pad_embed,
            req.config, ln.predictor_graph, ln.talker_graph, kw["max_new_tokens"], kw["min_new_tokens"],
            kw["temperature"], kw["top_k"], kw["top_p"], kw["do_sample"], kw["repetition_penalty"], use_graph=False)
        ln.req, ln.tn, ln.pn, ln.issued, ln.max_frames = req, tn, pn, 0, max_frames
        ln.t_arm, ln.prefill_ms = t0, (time.time() - t0) * 1000

    def _finish(self, ln: _Lane, n: int) -> Tuple[Any, Optional[torch.Tensor], Dict[str, float]]:
        codes = ln.engine.decode_codes(0, n) if n > 0 else None
        wall = time.time() - ln.t_arm
        timing = {"prefill_ms": ln.prefill_ms, "decode_s": max(wall - ln.prefill_ms / 1000, 0.0), "steps": n,
                  "ms_per_step": (1000 * wall / n) if n else 0.0, "steps_per_s": (n / wall) if wall > 0 else 0.0}
        rid = ln.req.rid
        ln.req, ln.tn, ln.pn = None, None, None
        return rid, codes, timing

    @torch.inference_mode()
    def run(self, requests: Iterable[BatchRequest], on_error: str = "raise"
            ) -> Iterator[Tuple[Any, Optional[torch.Tensor], Dict[str, Any]]]:
        """Yields ``(rid, codes LongTensor[T, 16] or None, timing)`` as utterances finish (not in request order).
        ``on_error="yield"``: a request that cannot be armed (prompt longer than ``max_seq_len``, nucleus sampling, ...)
        is reported as ``(rid, None, {"error": repr(exc), "steps": 0})`` and the other lanes keep going; the default
        re-raises, like the single-utterance entry points."""
        if on_error not in ("raise", "yield"):
            raise ValueError("on_error must be 'raise' or 'yield'")
        pending = deque(requests)
        free = deque(self.lanes)
        active: List[_Lane] = []
        while pending or active:
            while pending and free:                                   # admit at a frame boundary
                ln = free.popleft()
                req = pending.popleft()
                try:
                    self._arm(ln, req)
                except Exception as exc:
                    free.appendleft(ln)
                    if on_error == "raise":
                        raise
                    yield req.rid, None, {"error": repr(exc), "steps": 0}
                    continue
                if ln.max_frames <= 0:
                    yield self._finish(ln, 0)
                    free.append(ln)
                    continue
                active.append(ln)
            if not active:
                continue
            if self.use_graph and not self._captured:
                self.batch.graph_capture()
                self._captured = True
            # lock-step frames, never across a lane's noise-ring boundary (each lane refills its own rings)
            step = self.poll_every
            for ln in active:
                if ln.issued % NOISE_RING == 0:
                    _refill(ln.engine, ln.tn, ln.pn)
                step = min(step, NOISE_RING - ln.issued % NOISE_RING, max(ln.max_frames - ln.issued, 1))
            self.batch.frames(step)
            for ln in active:
                ln.issued += step
            still = []
            for ln in active:
                n, done = ln.engine.decode_poll()                     # first poll waits for the stream, the rest are free
                if done or ln.issued >= ln.max_frames:
                    yield self._finish(ln, n)
                    free.append(ln)
                else:
                    still.append(ln)
            active = still
