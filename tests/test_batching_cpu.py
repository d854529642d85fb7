"""CPU: the continuous-batching scheduler (fq3hip/batching.py) against scripted fake lanes -- admission at frame
boundaries, lock-step issue that never crosses a lane's noise-ring boundary, per-lane budgets, early EOS, lane re-use,
result routing by request id, error reporting."""
from types import SimpleNamespace

import pytest
import torch

import fq3hip.batching as Bt


class FakeLaneEngine:
    def __init__(self, log, idx):
        self.log, self.idx = log, idx
        self.dtype, self.device = torch.float32, torch.device("cpu")
        self.max_seq_len = 4096
        self.cfg = SimpleNamespace(num_code_groups=16, talker=SimpleNamespace(hidden_size=8, num_hidden_layers=1),
                                   predictor=SimpleNamespace(hidden_size=8, num_hidden_layers=1, vocab_size=32))
        self.frames, self.eos_after, self.budget, self.rid = 0, 10 ** 9, 0, None

    def set_predictor_sampling(self, **kw):
        pass

    def decode_poll(self):
        n = min(self.frames, self.eos_after, self.budget)
        return n, n >= self.eos_after

    def decode_codes(self, start, count):
        return torch.full((count, 16), float(self.rid)).long()


class FakeBatch:
    def __init__(self, engines):
        self.engines, self.calls, self.captured = engines, [], 0

    def graph_capture(self):
        self.captured += 1

    def frames(self, n):
        self.calls.append(n)
        for e in self.engines:
            e.frames += n


class LookAheadBatch(FakeBatch):
    """A batch object with the batch poll (fq3_batch_poll_async / _wait): the scheduler then queues the NEXT frames before it reads a
    poll.  The poll is a snapshot taken in stream order, i.e. when poll_async is called here."""

    def __init__(self, engines):
        super().__init__(engines)
        self.slots, self.trace = {}, []

    def frames(self, n):
        super().frames(n)
        self.trace.append(("frames", n))

    def poll_async(self, slot):
        assert slot not in self.slots, "a poll slot is re-used before it was read"
        self.slots[slot] = [e.decode_poll() for e in self.engines]
        self.trace.append(("poll", slot))

    def poll_wait(self, slot):
        snap = self.slots.pop(slot)
        self.trace.append(("wait", slot))
        return [n for n, _d in snap], [d for _n, d in snap]


@pytest.fixture(params=["sync", "lookahead"])
def sched(monkeypatch, request):
    log = []
    engines = [FakeLaneEngine(log, i) for i in range(3)]
    refills = []

    def fake_arm(talker, tie, tam, tth, tpe, config, pg, tg, max_new, min_new, temperature, top_k, top_p, do_sample, rp, use_graph):
        eng = tg.engine
        eng.frames, eng.budget, eng.eos_after, eng.rid = 0, int(max_new), config.eos_after, config.rid
        if getattr(config, "bad", False):
            raise RuntimeError("Input is too long")
        log.append(("arm", eng.idx, config.rid, top_p))
        assert use_graph is False
        return eng, torch.zeros(1), torch.zeros(1), int(max_new)

    monkeypatch.setattr(Bt, "_prefill_and_arm", fake_arm)
    monkeypatch.setattr(Bt, "_refill", lambda eng, tn, pn: refills.append((eng.idx, eng.frames)))
    monkeypatch.setattr(Bt, "TalkerGraph", lambda e: SimpleNamespace(engine=e))
    monkeypatch.setattr(Bt, "PredictorGraph", lambda e, **kw: SimpleNamespace(engine=e, top_p=kw.get("top_p", 1.0), **{k: v for k, v in kw.items() if k != "top_p"}))
    dec = Bt.BatchDecoder(engines, poll_every=8, batch_factory=FakeBatch if request.param == "sync" else LookAheadBatch)
    assert dec.lookahead == (0 if request.param == "sync" else 1)
    return dec, engines, log, refills


def _req(rid, max_new, eos_after=10 ** 9, **kw):
    cfg = SimpleNamespace(rid=rid, eos_after=eos_after)
    return Bt.BatchRequest(rid, None, torch.zeros(1, 4, 8), torch.ones(1, 4), torch.zeros(1, 2, 8), torch.zeros(1, 1, 8), cfg,
                           dict(max_new_tokens=max_new, **kw))


def test_more_requests_than_lanes_and_lane_reuse(sched):
    dec, engines, log, refills = sched
    reqs = [_req(0, 20), _req(1, 40, eos_after=13), _req(2, 24), _req(3, 8), _req(4, 16)]
    out = {rid: (codes, t) for rid, codes, t in dec.run(reqs)}
    assert set(out) == {0, 1, 2, 3, 4}
    assert [out[r][0].shape[0] for r in range(5)] == [20, 13, 24, 8, 16]           # budget / EOS per utterance
    assert all(int(out[r][0][0, 0]) == r for r in range(5))                        # routed by request id
    assert out[1][1]["steps"] == 13 and set(out[0][1]) == {"prefill_ms", "decode_s", "steps", "ms_per_step", "steps_per_s"}
    arms = [e for e in log if e[0] == "arm"]
    assert [a[2] for a in arms[:3]] == [0, 1, 2] and len(arms) == 5                # three lanes filled first, two re-used
    assert dec.batch.captured == 1                                                 # one graph for the whole run
    assert all(1 <= n <= 8 for n in dec.batch.calls)


def test_lock_step_never_crosses_a_noise_ring_boundary(sched):
    dec, engines, log, refills = sched
    list(dec.run([_req(0, 150), _req(1, 70)]))
    # every lane refills its own rings exactly at its frames 0, 64, 128, ...
    assert sorted(r for r in refills if r[0] == 0) == [(0, 0), (0, 64), (0, 128)]
    assert sorted(r for r in refills if r[0] == 1) == [(1, 0), (1, 64)]
    # second wave: a lane armed later has a different phase -> the step is cut at whichever boundary comes first
    dec2_calls = len(dec.batch.calls)
    list(dec.run([_req(5, 100)]))
    assert sum(dec.batch.calls[dec2_calls:]) == 100


def test_zero_budget_and_top_p_passes_through(sched):
    dec, engines, log, refills = sched
    out = list(dec.run([_req(0, 0), _req(1, 5)]))
    assert out[0][0] == 0 and out[0][1] is None and out[0][2]["steps"] == 0
    assert out[1][0] == 1 and out[1][1].shape[0] == 5
    # nucleus sampling is a per-lane policy now (the batch sampler kernels honour it): the request is armed like any other
    # and its top_p reaches the lane
    out = list(dec.run([_req(2, 5, top_p=0.9)]))
    assert out[0][1].shape[0] == 5
    assert [e for e in log if e[0] == "arm"][-1][3] == 0.9


def test_bad_request_can_be_reported_without_stopping_the_others(sched):
    dec, engines, log, refills = sched
    bad = _req(1, 5)
    bad.config.bad = True
    out = {rid: (c, t) for rid, c, t in dec.run([_req(0, 12), bad, _req(2, 9)], on_error="yield")}
    assert out[0][0].shape[0] == 12 and out[2][0].shape[0] == 9
    assert out[1][0] is None and "too long" in out[1][1]["error"] and out[1][1]["steps"] == 0
    with pytest.raises(ValueError):
        list(dec.run([], on_error="ignore"))


def test_staged_admission_prefills_ahead_and_reuses_spare_contexts(monkeypatch):
    """With spare contexts the scheduler prefills pending requests while lanes decode, admits them with kv_adopt + arm,
    recycles the spare context, and still routes / budgets every utterance like the direct path."""
    log = []

    class Eng(FakeLaneEngine):
        def kv_adopt(self, src, n_rows):
            log.append(("adopt", self.idx, src.idx, n_rows))

    lanes = [Eng(log, i) for i in range(2)]
    spares = [Eng(log, 10 + i) for i in range(2)]

    def fake_prefill(eng, tie, tam, config, min_new, temperature, top_k, top_p, do_sample):
        if getattr(config, "bad", False):
            raise RuntimeError("Input is too long")
        log.append(("prefill", eng.idx, config.rid, sum(e.frames for e in lanes)))
        return 7, torch.zeros(8), tie.shape[1], 0

    def fake_arm(talker, config, token, hidden, n_rows, tam, tth, tpe, pg, tg, max_new, min_new, temperature, top_k, top_p, do_sample, rp, use_graph, n_pad=None):
        eng = tg.engine
        cfg = eng.next_cfg
        eng.frames, eng.budget, eng.eos_after, eng.rid = 0, int(max_new), cfg.eos_after, cfg.rid
        log.append(("arm", eng.idx, cfg.rid))
        return eng, torch.zeros(1), torch.zeros(1), int(max_new)

    def fake_packed(engs, items):
        if any(getattr(it[2], "bad", False) for it in items):
            raise RuntimeError("Input is too long")           # a packed group fails as a whole; the scheduler then isolates the culprit
        log.append(("packed", [e.idx for e in engs], [it[2].rid for it in items]))
        return [fake_prefill(e, *it) for e, it in zip(engs, items)]

    monkeypatch.setattr(Bt, "_prefill_first_token", fake_prefill)
    monkeypatch.setattr(Bt, "_prefill_first_tokens_packed", fake_packed)
    monkeypatch.setattr(Bt, "_arm_decode", fake_arm)
    monkeypatch.setattr(Bt, "_refill", lambda eng, tn, pn: None)
    monkeypatch.setattr(Bt, "TalkerGraph", lambda e: SimpleNamespace(engine=e))
    monkeypatch.setattr(Bt, "PredictorGraph", lambda e, **kw: SimpleNamespace(engine=e, top_p=kw.get("top_p", 1.0)))
    dec = Bt.BatchDecoder(lanes, poll_every=8, batch_factory=FakeBatch, staging=spares, packed_prefill=True)
    orig_admit = dec._admit

    def admit(ln, st):
        ln.engine.next_cfg = st.req.config
        return orig_admit(ln, st)

    dec._admit = admit
    reqs = [_req(0, 24), _req(1, 40), _req(2, 16), _req(3, 8), _req(4, 8)]
    reqs[3].config.bad = True
    late = [_req(5, 8)]
    out = {rid: (c, t) for rid, c, t in dec.run(reqs, on_error="yield", source=lambda: late.pop() if late else None)}
    assert {r: (None if out[r][0] is None else out[r][0].shape[0]) for r in out} == {0: 24, 1: 40, 2: 16, 3: None, 4: 8, 5: 8}
    assert "too long" in out[3][1]["error"]
    prefills = [e for e in log if e[0] == "prefill"]
    assert [p[2] for p in prefills] == [0, 1, 2, 4, 5]                       # request order, the bad one never staged
    assert all(p[1] >= 10 for p in prefills)                                 # always into a spare context, never into a lane
    assert prefills[2][3] > 0 and prefills[3][3] > 0                         # requests 2 and 4 were prefilled while lanes were decoding
    packed = [e for e in log if e[0] == "packed"]
    assert packed[0][2] == [0, 1] and packed[0][1] == [10, 11]               # the first wave shares one pass over the weights
    assert all(len(p[2]) >= 2 for p in packed)
    adopts = [e for e in log if e[0] == "adopt"]
    assert len(adopts) == 5 and {a[2] for a in adopts} == {10, 11}           # spare contexts recycled
    assert [a[1] for a in adopts[:2]] == [0, 1] and all(a[3] == 4 for a in adopts)


def test_chunked_streaming_events(sched):
    """chunk_frames mode: whole chunks go out as they complete, the final event carries only the rest."""
    dec, engines, log, refills = sched
    events = list(dec.run([_req(0, 20), _req(1, 40, eos_after=13), _req(2, 16)], chunk_frames=8))
    per = {}
    for rid, codes, info in events:
        per.setdefault(rid, []).append((None if codes is None else codes.shape[0], info["is_final"], info["total_steps_so_far"]))
    assert per[0] == [(8, False, 8), (8, False, 16), (4, True, 20)]
    assert per[1] == [(8, False, 8), (5, True, 13)]                      # EOS inside the second chunk
    assert per[2] == [(8, False, 8), (8, True, 16)]                      # finished exactly at the polled boundary: no empty tail event needed
    finals = [info for _rid, _c, info in events if info["is_final"]]
    assert all("prefill_ms" in f and "steps" in f for f in finals)
    # default mode is unchanged: one event per utterance with all frames
    out = list(dec.run([_req(5, 12)]))
    assert len(out) == 1 and out[0][1].shape[0] == 12 and "is_final" not in out[0][2]


def test_abandoned_chunked_run_leaves_no_state_behind(sched):
    """A streaming consumer that stops early (generator closed after two chunks) must not make the next utterance on that lane
    lose its first frames: the partial-chunk counter is per tenant, not per lane."""
    dec, engines, log, refills = sched
    g = dec.run([_req(0, 40)], chunk_frames=8)
    first = [next(g), next(g)]
    assert [e[1].shape[0] for e in first] == [8, 8]
    g.close()
    events = list(dec.run([_req(1, 20)], chunk_frames=8))
    assert [(e[1].shape[0], e[2]["is_final"]) for e in events] == [(8, False), (8, False), (4, True)]
    assert sum(e[1].shape[0] for e in events) == 20
    # and a lane whose run died on an exception is clean for the next caller, too
    for ln in dec.lanes:
        ln.emitted = 16
    out = list(dec.run([_req(2, 12)]))
    assert out[0][1].shape[0] == 12


def test_late_failure_during_the_last_poll_is_still_reported(monkeypatch):
    """on_error='yield': a request that fails while being staged under the LAST frames of the last active lane (nothing left
    to decode afterwards) still produces its error event -- a server's reply queue would otherwise never be answered."""
    log = []

    class Eng(FakeLaneEngine):
        def kv_adopt(self, src, n_rows):
            pass

    lanes, spares = [Eng(log, 0)], [Eng(log, 10)]

    def fake_prefill(eng, tie, tam, config, min_new, temperature, top_k, top_p, do_sample):
        if getattr(config, "bad", False):
            raise RuntimeError("Input is too long")
        return 7, torch.zeros(8), tie.shape[1], 0

    def fake_arm(talker, config, token, hidden, n_rows, tam, tth, tpe, pg, tg, max_new, min_new, temperature, top_k, top_p, do_sample, rp, use_graph, n_pad=None):
        eng = tg.engine
        eng.frames, eng.budget, eng.eos_after, eng.rid = 0, int(max_new), 10 ** 9, config.rid
        return eng, torch.zeros(1), torch.zeros(1), int(max_new)

    monkeypatch.setattr(Bt, "_prefill_first_token", fake_prefill)
    monkeypatch.setattr(Bt, "_arm_decode", fake_arm)
    monkeypatch.setattr(Bt, "_refill", lambda eng, tn, pn: None)
    monkeypatch.setattr(Bt, "TalkerGraph", lambda e: SimpleNamespace(engine=e))
    monkeypatch.setattr(Bt, "PredictorGraph", lambda e, **kw: SimpleNamespace(engine=e, top_p=kw.get("top_p", 1.0)))
    dec = Bt.BatchDecoder(lanes, poll_every=8, batch_factory=FakeBatch, staging=spares)
    bad = _req(9, 8)
    bad.config.bad = True
    polls = [0]

    def source():                      # the bad request arrives just before request 0's only step (8 frames) is queued: it is
        polls[0] += 1                  # staged -- and fails -- under those frames, and request 0 finishes in the same iteration
        return bad if polls[0] == 2 else None

    out = {rid: (c, t) for rid, c, t in dec.run([_req(0, 8)], on_error="yield", source=source)}
    assert out[0][0].shape[0] == 8
    assert 9 in out and out[9][0] is None and "too long" in out[9][1]["error"]


def test_look_ahead_keeps_a_batch_of_frames_queued_but_never_past_a_known_finish(monkeypatch):
    """With the batch poll the scheduler reads poll k only after batch k + 1 has been queued (the GPU works while the host digests a
    poll) -- except when a lane's frame limit falls in the batch just queued: then it waits at once, and not one frame is queued for
    idle lanes.  An EOS (not predictable) costs at most the one batch already in flight."""
    log, refills = [], []
    engines = [FakeLaneEngine(log, i) for i in range(2)]

    def fake_arm(talker, tie, tam, tth, tpe, config, pg, tg, max_new, min_new, temperature, top_k, top_p, do_sample, rp, use_graph):
        eng = tg.engine
        eng.frames, eng.budget, eng.eos_after, eng.rid = 0, int(max_new), config.eos_after, config.rid
        return eng, torch.zeros(1), torch.zeros(1), int(max_new)

    monkeypatch.setattr(Bt, "_prefill_and_arm", fake_arm)
    monkeypatch.setattr(Bt, "_refill", lambda eng, tn, pn: None)
    monkeypatch.setattr(Bt, "TalkerGraph", lambda e: SimpleNamespace(engine=e))
    monkeypatch.setattr(Bt, "PredictorGraph", lambda e, **kw: SimpleNamespace(engine=e, **kw))
    dec = Bt.BatchDecoder(engines, poll_every=8, batch_factory=LookAheadBatch)
    out = {rid: c.shape[0] for rid, c, _t in dec.run([_req(0, 40), _req(1, 40)])}
    assert out == {0: 40, 1: 40} and sum(dec.batch.calls) == 40              # fixed lengths: no frame beyond the limit
    tr = dec.batch.trace
    # batch 1 is queued before poll 0 is read, batch 2 before poll 1, ...; the last batch (the limit falls in it) is read at once
    assert tr[:7] == [("frames", 8), ("poll", 0), ("frames", 8), ("poll", 1), ("wait", 0), ("frames", 8), ("poll", 2)]
    assert tr[-2:] == [("wait", 3), ("wait", 0)] and not dec.batch.slots
    # an utterance that ends by EOS at frame 13 of 200: found by the poll of the second batch, read while the third is in flight
    dec.batch.calls.clear()
    out = {rid: c.shape[0] for rid, c, _t in dec.run([_req(2, 200, eos_after=13)])}
    assert out == {2: 13} and sum(dec.batch.calls) == 24 and not dec.batch.slots
    # a lane re-armed while a batch is in flight: the poll of that batch still shows its predecessor and is ignored for it
    dec.batch.calls.clear()
    reqs = [_req(3, 200, eos_after=5), _req(4, 16), _req(5, 24)]
    out = {rid: c.shape[0] for rid, c, _t in dec.run(reqs)}
    assert out == {3: 5, 4: 16, 5: 24}


def test_staging_keeps_a_wide_scheduler_full(monkeypatch):
    """The requests staged per batch of frames scale with the lane count (max(2, lanes / 16)): with two per poll -- the rule up to
    round 3 -- a 48-lane scheduler refilled ~10 lanes per 40-frame wave and ran mostly empty (measured on the GPU at 128 lanes: 533x
    instead of 817x end to end).  Three waves of 48 fixed-length utterances must take little more than three waves of frames."""
    log = []

    class Eng(FakeLaneEngine):
        def kv_adopt(self, src, n_rows):
            pass

    n_lanes, frames, waves = 48, 40, 3
    lanes = [Eng(log, i) for i in range(n_lanes)]
    spares = [Eng(log, 100 + i) for i in range(n_lanes)]

    def fake_prefill(eng, tie, tam, config, min_new, temperature, top_k, top_p, do_sample):
        return 7, torch.zeros(8), tie.shape[1], 0

    def fake_arm(talker, config, token, hidden, n_rows, tam, tth, tpe, pg, tg, max_new, min_new, temperature, top_k, top_p, do_sample, rp, use_graph, n_pad=None):
        eng = tg.engine
        cfg = eng.next_cfg
        eng.frames, eng.budget, eng.eos_after, eng.rid = 0, int(max_new), cfg.eos_after, cfg.rid
        return eng, torch.zeros(1), torch.zeros(1), int(max_new)

    monkeypatch.setattr(Bt, "_prefill_first_token", fake_prefill)
    monkeypatch.setattr(Bt, "_prefill_first_tokens_packed", lambda engs, items: [fake_prefill(e, *it) for e, it in zip(engs, items)])
    monkeypatch.setattr(Bt, "_arm_decode", fake_arm)
    monkeypatch.setattr(Bt, "_refill", lambda eng, tn, pn: None)
    monkeypatch.setattr(Bt, "TalkerGraph", lambda e: SimpleNamespace(engine=e))
    monkeypatch.setattr(Bt, "PredictorGraph", lambda e, **kw: SimpleNamespace(engine=e, **kw))
    dec = Bt.BatchDecoder(lanes, poll_every=8, batch_factory=LookAheadBatch, staging=spares)
    orig_admit = dec._admit

    def admit(ln, st):
        ln.engine.next_cfg = st.req.config
        return orig_admit(ln, st)

    dec._admit = admit
    reqs = [_req(i, frames) for i in range(waves * n_lanes)]
    later = list(reversed(reqs[n_lanes:]))
    out = {rid: c.shape[0] for rid, c, _t in dec.run(reqs[:n_lanes], source=lambda: later.pop() if later else None)}
    assert len(out) == waves * n_lanes and set(out.values()) == {frames}
    total = sum(dec.batch.calls)
    assert total <= waves * frames + 2 * 8, total          # three full waves; with two requests per poll this took > 400 frames


def test_a_failed_admission_returns_the_lanes_blocks_and_lookahead_is_clamped(sched):
    """(round-4 advisor) A lane whose arming fails AFTER it took KV blocks (its own prefill, or the hand-over of a staged request, came
    before the failing step) has no tenant: the scheduler must cancel its loop state and return the blocks itself -- nothing else
    would.  And a look-ahead depth beyond the four poll slots is clamped to 3."""
    dec, engines, log, refills = sched
    held = {e.idx: 0 for e in engines}
    for e in engines:
        e.kv_blocks = (lambda e=e: held[e.idx])
        e.kv_release = (lambda keep=0, e=e: (held.__setitem__(e.idx, 0), log.append(("release", e.idx))))
        e.decode_cancel = (lambda e=e: log.append(("cancel", e.idx)))
    bad = _req(7, 16)
    bad.config.bad = True
    # the failing arm "took" three blocks before it raised
    orig = Bt._prefill_and_arm

    def arm_taking_blocks(talker, tie, tam, tth, tpe, config, pg, tg, *a, **kw):
        held[tg.engine.idx] = 3
        return orig(talker, tie, tam, tth, tpe, config, pg, tg, *a, **kw)

    Bt._prefill_and_arm = arm_taking_blocks
    try:
        dec.lookahead = 9 if dec.lookahead else 0
        out = {rid: (codes, t) for rid, codes, t in dec.run([_req(0, 12), bad, _req(2, 40)], on_error="yield")}
    finally:
        Bt._prefill_and_arm = orig
    assert out[7][0] is None and "too long" in out[7][1]["error"]
    assert out[0][0].shape[0] == 12 and out[2][0].shape[0] == 40
    failed_lane = [e[1] for e in log if e[0] == "cancel"]
    assert failed_lane and ("release", failed_lane[0]) in log                     # cancelled, then its blocks returned
    assert all(v == 0 for v in held.values())                                     # finished lanes returned theirs too


def test_deferred_first_tokens_are_read_group_by_group():
    """_stage_sync (round 6): the groups staged with defer=True are completed oldest first, and ``upto`` stops at the group that holds the
    stage the admission loop has reached -- the groups behind it stay deferred (their prefills keep the GPU busy meanwhile)."""
    dec = Bt.BatchDecoder.__new__(Bt.BatchDecoder)
    st = [Bt._Stage(engine=None) for _ in range(5)]
    for s in st:
        s.token = None
    reqs = [object() for _ in range(5)]
    mk = lambda idx: ([(st[i], reqs[i], None) for i in idx], [{} for _ in idx],
                      [(torch.tensor([100 + i]), f"hidden{i}", 7 + i, 0) for i in idx], 0.0, None, None)
    dec._unsynced = [mk([0, 1]), mk([2, 3]), mk([4])]
    dec._stage_sync(upto=st[0])
    assert [s.token for s in st] == [100, 101, None, None, None] and len(dec._unsynced) == 2
    dec._stage_sync(upto=st[3])
    assert [s.token for s in st] == [100, 101, 102, 103, None] and st[3].hidden == "hidden3" and st[3].n_rows == 10
    dec._stage_sync()                                                 # no stage named: every remaining group
    assert st[4].token == 104 and st[4].req is reqs[4] and dec._unsynced == []
    dec._stage_sync()                                                 # nothing deferred: a no-op
