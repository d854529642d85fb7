"""Side streams that really run beside the stream they are meant to overlap.

HIP multiplexes its streams onto a few hardware queues (four by default); two streams that land on the same queue execute in
submission order.  A vocoder stream that shares its queue with the decode stream does not overlap it: the first audio chunk then
waits for the NEXT chunk's frames, which were queued before it (measured on MI355X: p50 TTFA 28.6 -> 44-47 ms and RTF -10 % at the
1.7B shapes, for whichever of two model instances drew the unlucky stream).  ``concurrent_stream`` therefore PROBES: a spin on the
current stream, a trivial kernel on the candidate -- a candidate whose kernel finishes while the spin is still running has its own
queue.  The probe costs ~10 ms per candidate, once per component.  ``beside=`` names further streams the candidate must not share
a queue with (the batch scheduler's prefill stream and the lane groups' stream are probed against the vocoder's as well).
"""
from __future__ import annotations

from typing import Optional

import torch

_SPIN_CYCLES = 20_000_000          # ~10 ms


def _spin(stream):
    with torch.cuda.stream(stream):
        torch.cuda._sleep(_SPIN_CYCLES)


def concurrent_stream(device, priority: Optional[int] = None, tries: int = 8, beside=()) -> "torch.cuda.Stream":
    """A new stream on ``device`` that executes concurrently with the CURRENT stream and with every stream in ``beside`` (verified:
    all of them spin while the candidate's kernel completes).  There are only a few hardware queues: when no candidate runs beside
    all of them, one that at least runs beside the current stream is returned, else the last candidate tried."""
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    main = torch.cuda.current_stream(dev)
    others = [s for s in beside if s is not None and s != main]
    # the probe synchronises the device and creates streams: never while ANOTHER thread captures a hipGraph (several model instances
    # warming up side by side -- the capture would be invalidated: "operation failed due to a previous error during capture")
    from .engine import _CAPTURE_LOCK
    with _CAPTURE_LOCK:
        return _probe(dev, main, others, priority, tries)


def _probe(dev, main, others, priority, tries):
    cand, beside_main = None, None
    for _ in range(max(1, tries)):
        cand = torch.cuda.Stream(device=dev) if priority is None else torch.cuda.Stream(device=dev, priority=int(priority))
        try:
            torch.cuda.synchronize(dev)
            done = [torch.cuda.Event() for _ in range(1 + len(others))]
            done_cand = torch.cuda.Event()
            for st, ev in zip(others, done[1:]):
                _spin(st)
                ev.record(st)
            torch.cuda._sleep(_SPIN_CYCLES)
            done[0].record(main)
            with torch.cuda.stream(cand):
                x = torch.zeros(8, device=dev)
                x.add_(1)
                done_cand.record(cand)
            done_cand.synchronize()
            running = [not ev.query() for ev in done]
            torch.cuda.synchronize(dev)
        except Exception:               # no spin kernel on this build: take the stream as it is
            return cand
        if all(running):
            return cand
        if running[0] and beside_main is None:
            beside_main = cand
    return beside_main if beside_main is not None else cand


def cu_masked_stream(device, share: float) -> "torch.cuda.Stream":
    """A HIP stream whose kernels may only run on a SHARE of the compute units (``hipExtStreamCreateWithCUMask``; a measurement switch,
    ``FasterQwen3TTS.vocoder_cu_share``).  Idea: the vocoder's full-chip GEMM grids hold every CU for 100-400 us at a time and the
    latency-bound lock-step frames queue behind them; confined to part of the chip they leave the rest free at all times.  The enabled
    CUs are chosen by two parity bits of the CU index, so that every XCD / shader engine contributes the same share whichever way the
    driver maps mask bits to CUs.  ``share`` in {0.25, 0.5, 0.75}."""
    import ctypes as C
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    n_cu = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    words = (n_cu + 31) // 32
    quarters = max(1, min(3, int(round(float(share) * 4))))
    mask = (C.c_uint32 * words)()
    for b in range(n_cu):
        h0 = (b ^ (b >> 3) ^ (b >> 6)) & 1
        h1 = ((b >> 1) ^ (b >> 4) ^ (b >> 7)) & 1
        if h0 + 2 * h1 < quarters:
            mask[b // 32] |= 1 << (b % 32)
    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), mask)
    if rc != 0 or not st.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(st.value, device=dev)
