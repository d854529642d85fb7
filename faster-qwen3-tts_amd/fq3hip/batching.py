"""Continuous batching over ``Fq3Batch`` (``fq3_batch_*``): several utterances decode in lock-step over one weight
stream; a finished lane is re-armed with the next request at a frame boundary.  With ``staging`` contexts the next
requests are prefilled on a side stream WHILE the batch decodes, and a freed lane takes the staged prompt's KV over by a
block-table hand-over (``fq3_kv_adopt`` between contexts of one ``Fq3KvPool``: no KV row is copied), so admission costs the
running lanes a few tens of microseconds instead of a prefill.

KV memory follows the work, not the lane count: the contexts draw 64-key blocks from one pool; an idle lane or spare context
holds none, a staged request the blocks of its prompt + ``max_new_tokens`` (taken BEFORE its prefill is queued: a short pool
postpones the request instead of failing half-way), and a finished lane returns its blocks at once.

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
from typing import Any, Callable, Dict, Iterable, Iterator, List, Optional, Tuple

import torch

from ._lib import FQ3_ENOMEM
from .generate import (NOISE_RING, _arm_decode, _prefill_and_arm, _prefill_first_token, _prefill_first_tokens_launch,
                       _prefill_first_tokens_packed, _refill)
from .predictor_graph import PredictorGraph
from .talker_graph import TalkerGraph


MAX_LANES = 128         # kMaxLanes of csrc/batch_kernels.cuh (fq3_batch_create refuses more)


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
    emitted: int = 0             # frames already handed out as partial chunks (chunk_frames mode)
    max_frames: int = 0
    t_arm: float = 0.0
    prefill_ms: float = 0.0
    first_batch: int = 0         # index of the first batch of frames queued after this tenant was armed (polls of earlier batches show its predecessor)
    copied: Any = None           # event: the last side-stream read of this lane's code buffer (a new tenant is armed after it)


@dataclass
class _Stage:
    """A spare context a pending request is prefilled into (side stream) before a lane is free."""
    engine: Any
    req: Optional[BatchRequest] = None
    kw: Optional[Dict[str, Any]] = None
    token: int = 0
    hidden: Optional[torch.Tensor] = None
    n_rows: int = 0
    n_pad: int = 0               # left padding of the staged prompt (counted by the prefill)
    ready: Any = None            # event: prefill + first token done (side stream)
    released: Any = None         # event: the lane has copied the KV rows out (main stream)
    t0: float = 0.0
    prefill_ms: float = 0.0


class BatchDecoder:
    """Drives up to ``len(engines)`` lanes.  ``engines[1:]`` must share ``engines[0]``'s weights
    (``Fq3Engine(..., share=engines[0])``)."""

    def __init__(self, engines: List[Any], predictor_policy: Optional[Dict[str, Any]] = None, poll_every: int = 8,
                 use_graph: bool = True, batch_factory=None, staging: Optional[List[Any]] = None, packed_prefill: bool = True):
        if batch_factory is None:
            from .engine import Fq3Batch as batch_factory
        policy = predictor_policy or dict(do_sample=True, top_k=50, top_p=1.0, temperature=0.9)
        self.lanes = [_Lane(i, e, TalkerGraph(e), PredictorGraph(e, **policy)) for i, e in enumerate(engines)]
        self.batch = batch_factory(engines)
        self.poll_every = max(1, int(poll_every))
        self.more_in_poll = 0
        self.use_graph = use_graph
        self._captured = False
        # spare contexts (same weights, not part of the lock-step batch) for prefilling ahead of admission
        self.stages = [_Stage(e) for e in (staging or [])]
        self._side = None
        # other streams the host keeps busy while the batch decodes (the vocoder's): the prefill stream and the lane groups' stream are
        # probed for a hardware queue shared with none of them (fq3hip/streams.py)
        self.beside: list = []
        # LOOK-AHEAD (round 4): the next batch of frames is queued BEFORE the host waits for the previous batch's poll, so the GPU works
        # while the host digests a poll (finishes, chunk events, admissions, staged prefills: ~2 ms per poll of 64 lanes, ~5 ms with a
        # staging group; measured on MI355X: 13 % of a 64-lane end-to-end run was an idle decode queue, profiles/r04_e2e_timeline_*.txt).
        # The poll is one launch + one copy in stream order (fq3_batch_poll_async), read when its event has fired.  A batch that is
        # EXPECTED to finish a lane (its frame limit falls in it) is waited for at once -- queuing past a known finish would only burn
        # frames on idle lanes.  0 = wait for every batch right after queuing it (the behaviour up to round 3; what a batch object
        # without poll_async gets).
        self.lookahead = 1 if hasattr(self.batch, "poll_async") else 0
        self._copy = None           # side stream the codes of streaming chunks are read out on while the look-ahead frames run
        self._polls: set = set()    # poll slots queued and not read yet (an abandoned run leaves some behind)
        # every context this scheduler will ever PREFILL into gets its workspace NOW: a lazy device allocation inside a staged
        # prefill would land in the middle of the running lanes' decode.  With spare contexts the lanes never prefill (they only
        # adopt), so their workspaces would be dead memory (~100 MB each at 0.6B / 2048 slots, ~380 MB at 1.7B / 6144).
        for e in ([st.engine for st in self.stages] if self.stages else list(engines)):
            reserve = getattr(e, "prefill_reserve", None)
            if reserve is not None:
                reserve()
        # requests staged together share ONE pass over the weights (fq3_prefill_batch; parity-tested at the engine level,
        # tests/test_gpu_decode.py).  Measured with the workspaces reserved up front (profiles/r03_packed_prefill.txt, 0.6B shapes,
        # 200-row prompts): first-wave TTFA 82.5 -> 63.6 ms at 8 lanes and 125 -> 89 ms at 16, aggregate 153.5 -> 156x / 229 -> 234x.
        self.packed_prefill = bool(packed_prefill)
        # FIRST WAVE (round 5): how many requests are prepared, prefilled and armed before the first frame is queued (None = one per lane,
        # the behaviour up to round 4).  With many lanes and many simultaneous requests, preparing all of them first puts every prompt
        # build and every prefill in front of EVERYBODY's first chunk (128 lanes: first-wave TTFA 348 ms, of which ~250 ms is that
        # queue); a first wave of 32 starts decoding after 32 of them and the others join at the following frame boundaries, staged
        # under the running frames (`stage_limit`: up to a quarter of the lanes per poll).  MEASURED on MI355X (profiles/r05_ttfa_first_wave.txt):
        # it does not help -- the followers' prefills run beside the first wave's frames and stretch them (128 lanes, wave of 32: first
        # chunks at 305 .. 513 ms against 340 ms for all 128 at once; end to end 1005x against 1025x) -- so None stays the default and
        # the knob is a measurement switch.  What the latency of N simultaneous requests is made of: ~35 ms + 2.35 ms per request
        # (prefill 1.1, first-chunk vocoder 0.6 - 0.7, prompt build 0.33, arming 0.09).
        self.first_wave: Optional[int] = None

    def _group_streams(self):
        """When the batch was told to advance several lane groups concurrently (``fq3_batch_set_option("groups")``, a measurement switch:
        one chain is faster at every lane count), the further groups' streams are ours, so that they share a hardware queue neither
        with the decode stream nor with the vocoder / prefill streams."""
        eng = self.lanes[0].engine
        n_groups = getattr(self, "n_groups", None)                 # None: the library's choice (one chain)
        grouped = n_groups is not None and n_groups > 1
        if not grouped or not self._on_gpu(eng) or not hasattr(self.batch, "set_group_streams"):
            return
        from .streams import concurrent_stream
        if self._side is None and self.stages:
            self._side = concurrent_stream(eng.device, beside=self.beside)
        self._group_side = [concurrent_stream(eng.device, beside=list(self.beside) + [self._side])]
        for _ in range(2, n_groups or 2):                           # further groups (measurement): beside the decode stream at least
            self._group_side.append(concurrent_stream(eng.device, beside=self._group_side))
        self.batch.set_group_streams(self._group_side)

    def _more(self, queue: list):
        """Pop the next event of a poll's batch; ``self.more_in_poll`` tells the consumer (read it right after receiving the event) how
        many more events of the SAME poll follow: utterances that finish / chunks that complete together can then be vocoded by one
        batched codec launch set.  (An attribute, not a timing key: the timing dicts keep exactly the reference's keys.)"""
        ev = queue.pop(0)
        self.more_in_poll = len(queue)
        return ev

    def set_predictor_policy(self, **policy):
        for ln in self.lanes:
            for k, v in policy.items():
                setattr(ln.predictor_graph, k, v)

    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _kwargs(ln_pg, req: BatchRequest) -> Dict[str, Any]:
        kw = dict(max_new_tokens=2048, min_new_tokens=2, temperature=0.9, top_k=50, top_p=1.0, do_sample=True,
                  repetition_penalty=1.05)
        kw.update(req.gen_kwargs)
        # nucleus sampling (top_p < 1, sampling.py:57-65) is a per-lane policy of the loop state: the batch sampler kernels
        # branch to the LDS sorter for exactly the lanes that ask for it.  (Nothing here can fail: the values are validated by
        # fq3_decode_begin when the lane is armed.)
        return kw

    def _arm(self, ln: _Lane, req: BatchRequest):
        kw = self._kwargs(ln.predictor_graph, req)
        t0 = time.time()
        _eng, tn, pn, max_frames = _prefill_and_arm(
            req.talker, req.talker_input_embeds, req.attention_mask, req.trailing_text_hiddens, req.tts_pad_embed,
            req.config, ln.predictor_graph, ln.talker_graph, kw["max_new_tokens"], kw["min_new_tokens"],
            kw["temperature"], kw["top_k"], kw["top_p"], kw["do_sample"], kw["repetition_penalty"], use_graph=False)
        ln.req, ln.tn, ln.pn, ln.issued, ln.emitted, ln.max_frames = req, tn, pn, 0, 0, max_frames
        ln.t_arm, ln.prefill_ms = t0, (time.time() - t0) * 1000

    def _mark(self, engine):
        """Event after which the code tensors just read out are complete (recorded before further frames are queued)."""
        if not self._on_gpu(engine):
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(engine.device))
        return ev

    def _on_gpu(self, engine) -> bool:
        return torch.cuda.is_available() and getattr(engine.device, "type", "cpu") == "cuda"

    def _stage(self, st: _Stage, req: BatchRequest, req_ready=None):
        """Prefill + first token of ``req`` into the spare context, on the side stream (the host waits for the token, the
        main stream -- with lock-step frames already queued -- does not)."""
        self._stage_many([(st, req, req_ready)])

    def _stage_sync(self, upto=None):
        """The first tokens of the groups staged with ``defer=True``, oldest first: every group's tokens were copied to pinned host
        memory right behind its prefill (stream order), so reading group g waits for ITS prefill only -- the groups behind it keep the
        GPU busy while the host arms group g's lanes (round 6; until then ONE copy fetched all first tokens, and the GPU idled through
        the arming of a whole wave: 14 ms at 128 lanes).  ``upto``: stop once this stage's fields are complete (None = all groups)."""
        pend = getattr(self, "_unsynced", [])
        while pend:
            group, kws, res, t0, host, done = pend.pop(0)
            if done is not None:
                done.synchronize()                                    # this group's prefill + its token copy; later groups still run
                toks = host.tolist()
            else:
                toks = torch.stack([r[0].reshape(()) for r in res]).cpu().tolist()
            ms = (time.time() - t0) * 1000
            hit = False
            for (st, req, _ev), kw, (_tok, hidden, n_rows, n_pad), tok in zip(group, kws, res, toks):
                st.req, st.kw, st.token, st.hidden, st.n_rows, st.n_pad = req, kw, int(tok), hidden, n_rows, int(n_pad)
                st.t0, st.prefill_ms = t0, ms
                hit = hit or st is upto
            if hit:
                break

    def _stage_many(self, group, defer: bool = False):
        """``[(spare context, request, ready event), ...]``: ONE packed prefill for the whole group when it has more than one
        member (``fq3_prefill_batch``: the layer weights are read once, not once per request).  ``defer`` (the pipelined first wave,
        GPU engines with packed prefill only): nothing is waited for -- the left-padding counts of the whole group cost one device
        round trip up front, the first tokens stay on the device until :meth:`_stage_sync` -- so the host goes on building the next
        prompts while these prefills run."""
        kws = [self._kwargs(self.lanes[0].predictor_graph, req) for _st, req, _ev in group]
        defer = bool(defer) and self.packed_prefill and self._on_gpu(group[0][0].engine)
        t0 = time.time()
        items = [(req.talker_input_embeds, req.attention_mask, req.config, kw["min_new_tokens"], kw["temperature"], kw["top_k"],
                  kw["top_p"], kw["do_sample"]) for (_st, req, _ev), kw in zip(group, kws)]
        engines = [st.engine for st, _req, _ev in group]
        taken = []
        host_toks = None

        def reserve_all():
            # the KV blocks of every member's whole utterance (prompt + max_new_tokens + 1 slots, capped at max_seq_len) are taken
            # BEFORE anything is queued: a short pool raises Fq3Error(FQ3_ENOMEM) here with every context as it was.  Called on the
            # stream the prefill runs on: the block-table entries are written by a launch on the CURRENT stream, and the prefill
            # kernels read them.
            try:
                for e, it, kw in zip(engines, items, kws):
                    reserve = getattr(e, "kv_reserve", None)
                    if reserve is not None:
                        reserve(int(it[0].shape[1]) + int(kw["max_new_tokens"]) + 1)
                        taken.append(e)
            except Exception:
                for e in taken:
                    e.kv_release()
                del taken[:]
                raise

        def run_deferred():
            # left padding of every prompt: one stack, one copy (the per-request int() of the synchronous path is a device wait each)
            masks = [it[1] for it in items]
            from .generate import noted_n_pad
            known = [0 if m is None else noted_n_pad(m) for m in masks]
            if all(k is not None for k in known):
                cnt = [int(k) for k in known]                         # the prompt builder noted them: no device round trip at all
            else:
                cnt = torch.stack([(m[0] == 0).sum() if m is not None else torch.zeros((), dtype=torch.long, device=engines[0].device) for m in masks]).cpu().tolist()
            out, lo, rows = [], 0, 0
            cap = int(getattr(engines[0], "max_seq_len", 1 << 30))
            for i, it in enumerate(items):
                n = int(it[0].shape[1])
                if i > lo and rows + n > cap:
                    out += _prefill_first_tokens_launch(engines[lo:i], items[lo:i], cnt[lo:i])
                    lo, rows = i, 0
                rows += n
            out += _prefill_first_tokens_launch(engines[lo:], items[lo:], cnt[lo:])
            return out

        def run():
            if defer:
                return run_deferred()
            if len(group) == 1 or not self.packed_prefill:
                return [_prefill_first_token(e, *it) for e, it in zip(engines, items)]
            # packed prefills share the leading context's workspace (max_seq_len rows): split the group where the rows run out
            out, lo, rows = [], 0, 0
            cap = int(getattr(engines[0], "max_seq_len", 1 << 30))
            for i, it in enumerate(items):
                n = int(it[0].shape[1])
                if i > lo and rows + n > cap:
                    out += (_prefill_first_tokens_packed(engines[lo:i], items[lo:i]) if i - lo > 1
                            else [_prefill_first_token(engines[lo], *items[lo])])
                    lo, rows = i, 0
                rows += n
            out += (_prefill_first_tokens_packed(engines[lo:], items[lo:]) if len(items) - lo > 1
                    else [_prefill_first_token(engines[lo], *items[lo])])
            return out

        try:
            if self._on_gpu(engines[0]):
                dev = engines[0].device
                if self._side is None:
                    from .streams import concurrent_stream
                    self._side = concurrent_stream(dev, beside=self.beside)     # verified to run beside the decode (and vocoder) stream
                for _st, _req, ev in group:
                    if ev is not None:
                        self._side.wait_event(ev)                           # the request's tensors were produced on the main stream
                    else:
                        self._side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(self._side):
                    for st, _req, _ev in group:
                        if st.released is not None:
                            self._side.wait_event(st.released)              # the previous tenant's hand-over has been queued
                    reserve_all()
                    res = run()
                    if defer:
                        # the group's first tokens go to pinned host memory right behind its prefill: no wait here
                        flat = torch.stack([r[0].reshape(()) for r in res])
                        host_toks = torch.empty(flat.shape, dtype=flat.dtype, pin_memory=True)
                        host_toks.copy_(flat, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(self._side)
                for st, _req, _ev in group:
                    st.ready = done
            else:
                reserve_all()
                res = run()
        except Exception:
            for e in taken:                                                 # a failed prefill keeps no blocks
                e.kv_release()
            raise
        if defer:
            for (st, req, _ev), kw in zip(group, kws):
                st.req, st.kw, st.token = req, kw, None               # (the stage is taken; token / hidden follow in _stage_sync)
            self._unsynced = getattr(self, "_unsynced", []) + [(group, kws, res, t0, host_toks, group[0][0].ready if host_toks is not None else None)]
            return
        ms = (time.time() - t0) * 1000
        for (st, req, _ev), kw, (token, hidden, n_rows, n_pad) in zip(group, kws, res):
            st.req, st.kw, st.token, st.hidden, st.n_rows, st.n_pad = req, kw, token, hidden, n_rows, int(n_pad)
            st.t0, st.prefill_ms = t0, ms

    def _admit(self, ln: _Lane, st: _Stage):
        """Hand a staged request to a free lane at a frame boundary: the block-table hand-over (one tiny launch; a block-wise copy
        if the two contexts do not share a pool) + the arm kernel."""
        req, kw = st.req, st.kw
        gpu = self._on_gpu(ln.engine)
        if gpu:
            main = torch.cuda.current_stream(ln.engine.device)
            main.wait_event(st.ready)
            st.hidden.record_stream(main)
        ln.engine.kv_adopt(st.engine, st.n_rows)
        if gpu:
            st.released = torch.cuda.Event()
            st.released.record(main)
        _eng, tn, pn, max_frames = _arm_decode(
            req.talker, req.config, st.token, st.hidden, st.n_rows, req.attention_mask, req.trailing_text_hiddens, req.tts_pad_embed,
            ln.predictor_graph, ln.talker_graph, kw["max_new_tokens"], kw["min_new_tokens"], kw["temperature"], kw["top_k"],
            kw["top_p"], kw["do_sample"], kw["repetition_penalty"], use_graph=False, n_pad=st.n_pad)
        ln.req, ln.tn, ln.pn, ln.issued, ln.emitted, ln.max_frames = req, tn, pn, 0, 0, max_frames
        ln.t_arm, ln.prefill_ms = st.t0, st.prefill_ms
        st.req, st.kw, st.hidden = None, None, None

    def _drop_blocks(self, x, cancel: bool = False):
        """Return whatever KV blocks lane / stage ``x`` owns to the pool (a lane's device loop is cancelled first when asked: a lane that
        was armed must not append to blocks it no longer owns)."""
        eng = x.engine
        try:
            if cancel and getattr(eng, "decode_cancel", None) is not None:
                eng.decode_cancel()
            if getattr(eng, "kv_release", None) is not None:
                eng.kv_release()
        except Exception:
            pass

    def _fetch(self, ln: _Lane, first: int, count: int, ev_done=None):
        """``(codes LongTensor[count, 16] from frame `first`, event after which they are complete)``.  With look-ahead frames in flight
        (``ev_done``: the event behind the batch that produced them) the read-out runs on the copy stream, beside those frames."""
        eng = ln.engine
        if self._copy is None or ev_done is None:
            return eng.decode_codes(first, count), self._mark(eng)
        with torch.cuda.stream(self._copy):
            self._copy.wait_event(ev_done)
            codes = eng.decode_codes(first, count)
            ev = torch.cuda.Event()
            ev.record(self._copy)
        ln.copied = ev
        return codes, ev

    def _finish(self, ln: _Lane, n: int, chunked: bool = False, ev_done=None) -> Tuple[Any, Optional[torch.Tensor], Dict[str, float]]:
        first = ln.emitted if chunked else 0
        ready_ev = None
        if n > first:
            if chunked:
                codes, ready_ev = self._fetch(ln, first, n - first, ev_done)
            else:
                codes = ln.engine.decode_codes(first, n - first)
        elif chunked and n > 0:                      # every frame already went out as a partial chunk
            codes = torch.empty(0, ln.engine.cfg.num_code_groups, dtype=torch.long, device=ln.engine.device)
        else:
            codes = None
        wall = time.time() - ln.t_arm
        timing = {"prefill_ms": ln.prefill_ms, "decode_s": max(wall - ln.prefill_ms / 1000, 0.0), "steps": n,
                  "ms_per_step": (1000 * wall / n) if n else 0.0, "steps_per_s": (n / wall) if wall > 0 else 0.0}
        if chunked:
            timing.update(is_final=True, total_steps_so_far=n)
            if codes is not None:
                timing["codes_ready_event"] = ready_ev if ready_ev is not None else self._mark(ln.engine)
        rid = ln.req.rid
        ln.req, ln.tn, ln.pn, ln.emitted = None, None, None, 0
        # the lane's KV blocks go back to the pool now: the poll that found it finished waited for its last frame, and a done lane never
        # touches the cache again -- not in look-ahead frames still in flight either (done-lane guard, csrc/batch_kernels.cuh)
        release = getattr(ln.engine, "kv_release", None)
        if release is not None:
            release()
        return rid, codes, timing

    @torch.inference_mode()
    def run(self, requests: Iterable[BatchRequest], on_error: str = "raise",
            source: Optional[Callable[[], Optional[BatchRequest]]] = None, chunk_frames: Optional[int] = None
            ) -> Iterator[Tuple[Any, Optional[torch.Tensor], Dict[str, Any]]]:
        """Yields ``(rid, codes LongTensor[T, 16] or None, timing)`` as utterances finish (not in request order).
        ``on_error="yield"``: a request that cannot be armed (prompt longer than ``max_seq_len``, ...)
        is reported as ``(rid, None, {"error": repr(exc), "steps": 0})`` and the other lanes keep going; the default
        re-raises, like the single-utterance entry points.  ``source``: polled without blocking at every frame boundary
        for requests that arrived after the call (``None`` = nothing waiting): a server's inbox.
        ``chunk_frames``: streaming mode -- besides the final event every utterance yields partial events
        ``(rid, codes of the next whole chunks, {"is_final": False, "total_steps_so_far": n})`` as its frames complete, and
        its final event (``"is_final": True``) carries only the frames not handed out before (``None`` when the utterance
        produced no frame at all, an empty tensor when everything was already handed out)."""
        chunked = chunk_frames is not None and int(chunk_frames) > 0
        if on_error not in ("raise", "yield"):
            raise ValueError("on_error must be 'raise' or 'yield'")
        gpu = self._on_gpu(self.lanes[0].engine)

        def stamped(req):
            """(request, event after which its tensors are ready): recorded on the main stream when the request arrives, so
            a side-stream prefill waits for THAT, not for the lock-step frames queued later."""
            ev = None
            if gpu and self.stages:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.lanes[0].engine.device))
            return req, ev

        # a previous run() may have been abandoned mid-utterance (a streaming consumer that stopped early, an exception in the
        # caller): no lane carries a tenant, a partial-chunk counter or a staged request over into this one
        def owns(x):
            blocks = getattr(x.engine, "kv_blocks", None)
            return blocks is not None and blocks() > 0
        abandoned = [x for x in list(self.lanes) + list(self.stages) if x.req is not None or owns(x)]
        if abandoned:
            for x in abandoned:
                cancel = getattr(x.engine, "decode_cancel", None)
                if cancel is not None and isinstance(x, _Lane):
                    cancel()                                  # the device loop state stops at once instead of decoding to its own EOS
            if gpu:
                torch.cuda.current_stream(self.lanes[0].engine.device).synchronize()       # its queued frames are over: the blocks may go
                if self._side is not None:
                    self._side.synchronize()
            for x in abandoned:
                release = getattr(x.engine, "kv_release", None)
                if release is not None:
                    release()
        for slot in sorted(self._polls):                  # polls an abandoned run queued and never read
            self.batch.poll_wait(slot)
        self._polls.clear()
        self._unsynced = []                               # (first tokens an abandoned run left on the device: their stages are reset below)
        for ln in self.lanes:
            ln.req, ln.tn, ln.pn, ln.issued, ln.emitted = None, None, None, 0, 0
        for st in self.stages:
            st.req, st.kw, st.hidden = None, None, None
        pending = deque(stamped(r) for r in requests)
        free = deque(self.lanes)
        active: List[_Lane] = []
        idle = deque(self.stages)
        ready: deque = deque()
        failed: List[Tuple[Any, Dict[str, Any]]] = []
        outbox: List[Tuple[Any, Any, Dict[str, Any]]] = []            # streaming-mode events waiting for the next frames to be queued
        # host-time breakdown of this run (seconds; development aid, read by tools/batch_e2e_bench.py): where the scheduler's thread spends
        # its time between two batches of frames
        prof = self.host_s = {"stage": 0.0, "admit": 0.0, "queue": 0.0, "poll_wait": 0.0, "digest": 0.0, "consumer": 0.0}
        clock = time.perf_counter

        # Requests taken from the source / staged per batch of frames while lanes decode.  A floor that keeps long utterances flowing
        # (lanes / 16 per poll refills every lane within 16 polls = 128 frames; two per poll -- the rule up to round 3 -- starves a
        # scheduler of more than ~50 lanes at 200-frame utterances: 128 lanes ran half empty), topped up by DEMAND: lanes that are
        # free now, lanes whose frame limit falls within the next four polls, and the recent rate of early (EOS) finishes, minus what
        # is staged already -- at most a quarter of the lanes per poll, so that the host work of one poll stays inside the batch of
        # frames queued ahead.  Short utterances (a 48-lane scheduler of 40-frame utterances turns over every 5 polls) are what
        # needs the top-up.
        per_poll = max(2, (len(self.lanes) + 15) // 16)
        per_poll_cap = max(per_poll, len(self.lanes) // 4)
        finish_rate = [0.0]                                             # finishes per poll, exponentially averaged

        def stage_limit():
            """Requests to stage in this poll (the source is asked for what `pending` does not hold already)."""
            horizon = 4 * self.poll_every
            soon = sum(1 for ln in active if ln.max_frames - ln.issued <= horizon)
            need = len(free) + soon + int(finish_rate[0] + 0.999) - len(ready)
            return max(per_poll, min(need, per_poll_cap))

        def pull(cap: Optional[int] = None, on_main: bool = False):
            # while lanes decode, at most two new requests per frame boundary: whatever the source does to produce one (a
            # model's prompt build, say) runs on the host between two batches of queued frames.  Before anything decodes: only
            # what the first wave can take (one request per lane) -- every further prompt built now would delay the first frame
            wave = len(self.lanes) if self.first_wave is None else max(1, min(int(self.first_wave), len(self.lanes)))
            budget = max(0, wave - len(pending) - len(ready)) if not active else max(0, stage_limit() - len(pending))
            if cap is not None:
                budget = min(budget, int(cap))
            while source is not None and budget > 0 and len(pending) < len(self.lanes) + len(self.stages):
                # with spare contexts the source runs under the PREFILL stream: whatever device work it does to produce a request (a
                # model's prompt build) neither queues behind the lock-step frames in flight nor -- where it waits for a value --
                # makes the host wait for them; the staged prefill that consumes it is on that stream anyway
                # (first wave, `on_main`: nothing decodes, the MAIN stream is idle -- the builder's host-to-device uploads are synchronous
                # copies, and on the prefill stream each would wait for the packed prefill queued in front of it)
                if feed_stream is not None and not on_main:
                    with torch.cuda.stream(feed_stream):
                        r = source()
                        r = None if r is None else stamped(r)
                else:
                    r = source()
                    r = None if r is None else stamped(r)
                if r is None:
                    break
                pending.append(r)
                budget -= 1

        def stage_ahead(limit: int = 1 << 30, defer: bool = False) -> int:
            # while lanes decode, only a couple of prefills per batch of queued frames: the host waits for the staged requests'
            # first tokens, and the main stream must not run dry meanwhile.  Requests staged together are prefilled together.
            group = []
            while pending and idle and limit > 0:
                limit -= 1
                st, (req, ev) = idle.popleft(), pending.popleft()
                group.append((st, req, ev))
            if not group:
                return 0
            try:
                self._stage_many(group, defer=defer)
                ready.extend(st for st, _r, _e in group)
                return len(group)
            except Exception as exc:
                if getattr(exc, "code", None) == FQ3_ENOMEM and (active or ready):
                    # the KV pool is short right now: lanes that finish give blocks back -- keep the requests (in order) for later
                    for st, req, ev in reversed(group):
                        idle.appendleft(st)
                        pending.appendleft((req, ev))
                    return 0
                if len(group) == 1:
                    st, req, _ev = group[0]
                    idle.appendleft(st)
                    if on_error == "raise":
                        raise
                    import sys
                    failed.append((req.rid, {"error": repr(sys.exc_info()[1]), "steps": 0}))
                    return 0
            # a packed group failed (one prompt too long, the pool too short for all of them, ...): stage its members one by one so
            # that only the culprit fails
            postponed = []
            for st, req, ev in group:
                try:
                    if postponed:
                        raise postponed[0][3]                         # keep the order: nothing overtakes a postponed request
                    self._stage_many([(st, req, ev)])
                    ready.append(st)
                except Exception as exc:
                    if getattr(exc, "code", None) == FQ3_ENOMEM and (active or ready):
                        postponed.append((st, req, ev, exc))
                        continue
                    idle.appendleft(st)
                    if on_error == "raise":
                        raise
                    failed.append((req.rid, {"error": repr(exc), "steps": 0}))
            for st, req, ev, _exc in reversed(postponed):
                idle.appendleft(st)
                pending.appendleft((req, ev))
            return len(group) - len(postponed)

        # (four poll slots, indexed by the batch number modulo 4: with `depth` batches unread and one being queued, depth <= 3 keeps a slot
        # from being re-armed before its poll has been read)
        depth = max(0, min(int(self.lookahead), 3)) if hasattr(self.batch, "poll_async") else 0
        feed_stream = None
        if gpu and self.stages:
            if self._side is None:
                from .streams import concurrent_stream
                self._side = concurrent_stream(self.lanes[0].engine.device, beside=self.beside)    # verified to run beside the decode (and vocoder) stream
            feed_stream = self._side
        if chunked and depth and gpu and self._copy is None:
            # streaming: the codes of a chunk are read out beside the look-ahead frames, not behind them -- on the consumer's own stream
            # (the vocoder's, `beside[0]`: hardware queues are few), else on a stream probed for a queue of its own
            if self.beside:
                self._copy = self.beside[0]
            else:
                from .streams import concurrent_stream
                dev = self.lanes[0].engine.device
                if self._side is None and self.stages:
                    self._side = concurrent_stream(dev, beside=self.beside)
                self._copy = concurrent_stream(dev, beside=[self._side])
        inflight: deque = deque()           # batches of frames whose poll has not been read yet: (batch index, poll slot | None, event | None)
        batch_no = 0

        def arm_after_copy(ln):
            # a side-stream read of this lane's code buffer (streaming look-ahead) must be over before a new tenant's frames overwrite it
            if ln.copied is not None:
                torch.cuda.current_stream(ln.engine.device).wait_event(ln.copied)
                ln.copied = None

        def digest(entry):
            """Read one batch's poll: chunk events and finishes of the lanes that were armed before it was queued."""
            nonlocal active
            bno, slot, ev_done = entry
            if slot is not None:
                t_ = clock()
                n_all, d_all = self.batch.poll_wait(slot)               # waits for THIS batch's frames only: later ones keep running
                prof["poll_wait"] += clock() - t_
                self._polls.discard(slot)
            still = []
            for ln in active:
                if ln.first_batch > bno:                                # armed after this batch was queued: the poll shows its predecessor
                    still.append(ln)
                    continue
                n, done = (n_all[ln.index], d_all[ln.index]) if slot is not None else ln.engine.decode_poll()
                fin = done or n >= ln.max_frames
                if chunked:
                    # whole chunks, one event each (a poll can complete several); a finished utterance's last event carries
                    # the remainder (at most one chunk) and the timing
                    c = int(chunk_frames)
                    upto = n if fin else (n // c) * c
                    while (upto - ln.emitted > c) if fin else (upto - ln.emitted >= c):
                        codes, ready_ev = self._fetch(ln, ln.emitted, c, ev_done)
                        ln.emitted += c
                        outbox.append((ln.req.rid, codes, {"is_final": False, "total_steps_so_far": ln.emitted,
                                                           "codes_ready_event": ready_ev}))
                if fin:
                    # like the chunk events, a finish goes out only once the next frames are queued: whatever the consumer does with
                    # it (a vocoder launch set for a whole wave of utterances is ~15 ms of host time) then runs beside the decode of
                    # the lanes' next tenants instead of in front of it
                    outbox.append(self._finish(ln, n, chunked, ev_done))
                    free.append(ln)
                else:
                    still.append(ln)
            finish_rate[0] = 0.5 * finish_rate[0] + 0.5 * (len(active) - len(still))
            active = still

        # FIRST WAVE, pipelined (round 5): nothing decodes yet, so the critical path to the first frame is "build the prompts (host)" +
        # "prefill them (GPU)".  They used to run one after the other (128 lanes: 39 ms + 96 ms); now the prompts arrive in slices and
        # every slice's packed prefill is QUEUED without waiting for its first tokens (`defer`), so the host builds slice k + 1 while the
        # GPU prefills slice k; one copy fetches all first tokens before the lanes are armed.
        slice_n = max(1, int(getattr(self, "first_slice", 16)))
        while True:
            first_wave_phase = bool(self.stages) and not active and not ready and not inflight
            pull(cap=slice_n if first_wave_phase else None, on_main=first_wave_phase)
            if not (pending or active or ready or failed or inflight):
                break
            if self.stages and not active and not ready:
                t_ = clock()
                try:
                    while pending:                                    # nothing is decoding: nothing to overlap with but the prompt builds
                        if stage_ahead(defer=True) == 0:
                            break
                        pull(cap=slice_n, on_main=True)
                except BaseException:
                    # the deferred stages hold a request but not yet its first token / hidden state: complete them when the interleaved
                    # pull() (the caller's source) raised, so that no stage is left half-filled for the cleanup to find
                    self._stage_sync()
                    raise
                # (no wait here: the admission loop below reads each group's first tokens when it reaches the group's first stage, and
                # arms its lanes while the groups behind it are still being prefilled)
                prof["stage"] += clock() - t_
            while failed:
                rid, info = failed.pop(0)
                self.more_in_poll = 0
                yield rid, None, info
            t_admit = clock()
            while free and (ready or (pending and not self.stages)):  # admit at a frame boundary
                ln = free.popleft()
                try:
                    arm_after_copy(ln)
                    if ready:
                        st = ready.popleft()
                        rid = st.req.rid
                        try:
                            if st.token is None:
                                self._stage_sync(upto=st)             # a deferred group: its first tokens (and only its) are waited for
                            self._admit(ln, st)
                        finally:
                            idle.append(st)
                    else:
                        req, _ev = pending.popleft()
                        rid = req.rid
                        self._arm(ln, req)
                    ln.first_batch = batch_no                         # the frames queued from now on are this tenant's
                except Exception as exc:
                    free.appendleft(ln)
                    # the lane may already own blocks (the hand-over of a staged request, or its own prefill, came before the step that
                    # failed -- fq3_decode_begin rejecting a sampling argument, say): it has no tenant, so nothing else would return them
                    self._drop_blocks(ln, cancel=True)
                    if on_error == "raise":
                        raise
                    self.more_in_poll = 0
                    yield rid, None, {"error": repr(exc), "steps": 0}
                    continue
                if ln.max_frames <= 0:
                    self.more_in_poll = 0
                    yield self._finish(ln, 0, chunked)
                    free.append(ln)
                    continue
                active.append(ln)
            prof["admit"] += clock() - t_admit
            if not active and not inflight:
                continue
            if self.use_graph and not self._captured:
                self._group_streams()
                self.batch.graph_capture()
                self._captured = True
            # lock-step frames, never across a lane's noise-ring boundary (each lane refills its own rings); lanes whose frame limit is
            # already covered by the frames in flight need no more (they wait for their poll)
            need = [ln for ln in active if ln.issued < ln.max_frames]
            t_ = clock()
            if need:
                step = self.poll_every
                for ln in need:
                    if ln.issued % NOISE_RING == 0:
                        _refill(ln.engine, ln.tn, ln.pn)
                    step = min(step, NOISE_RING - ln.issued % NOISE_RING, max(ln.max_frames - ln.issued, 1))
                pull()                                                # stamp new arrivals before the frames are queued
                self.batch.frames(step)
                for ln in need:
                    ln.issued += step
                slot = ev_done = None
                if depth:
                    slot = batch_no % 4
                    self.batch.poll_async(slot)                       # in stream order: behind these frames, in front of the next batch
                    self._polls.add(slot)
                    if self._copy is not None and chunked:
                        ev_done = self._mark(self.lanes[0].engine)
                inflight.append((batch_no, slot, ev_done))
                batch_no += 1
            prof["queue"] += clock() - t_
            # the chunks and finishes found by the previous poll go out only now, with the next frames already queued, so
            # that whatever the consumer does with them (vocoding) overlaps the decode instead of stalling it
            t_ = clock()
            while outbox:
                yield self._more(outbox)
            prof["consumer"] += clock() - t_
            if self.stages and active:
                t_ = clock()
                stage_ahead(limit=stage_limit())      # prefills fly under the frames queued above
                prof["stage"] += clock() - t_
            # wait for the oldest batch; for ALL of them when a lane's frame limit falls in the newest (a finish is expected: queuing
            # past it would burn frames on idle lanes) or when nothing more could be queued
            expect = any(ln.issued >= ln.max_frames for ln in active)
            while inflight and (len(inflight) > depth or expect or not need):
                t_, w_ = clock(), prof["poll_wait"]
                digest(inflight.popleft())
                prof["digest"] += clock() - t_ - (prof["poll_wait"] - w_)
                if not active:
                    while inflight:                                   # frames queued past the last finish: nothing left to read in them
                        _b, slot, _e = inflight.popleft()
                        if slot is not None:
                            self.batch.poll_wait(slot)
                            self._polls.discard(slot)
            if inflight or (not active and not ready and not pending):   # frames are running (or nothing is left to overlap with)
                while outbox:
                    yield self._more(outbox)
        while outbox:
            yield self._more(outbox)
        while failed:                                                 # belt and braces: no error event is ever dropped
            rid, info = failed.pop(0)
            self.more_in_poll = 0
            yield rid, None, info
