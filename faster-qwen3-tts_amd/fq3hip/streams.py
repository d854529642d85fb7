"""Side streams that really run beside the stream they are meant to overlap.

HIP multiplexes its streams onto a few hardware queues (four by default); two streams that land on the same queue execute in
submission order.  A vocoder stream that shares its queue with the decode stream does not overlap it: the first audio chunk then
waits for the NEXT chunk's frames, which were queued before it (measured on MI355X: p50 TTFA 28.6 -> 44-47 ms and RTF -10 % at the
1.7B shapes, for whichever of two model instances drew the unlucky stream).  ``concurrent_stream`` therefore PROBES: a spin on the
current stream, a trivial kernel on the candidate -- a candidate whose kernel finishes while the spin is still running has its own
queue.  The probe costs ~10 ms per candidate, once per component.
"""
from __future__ import annotations

from typing import Optional

import torch

_SPIN_CYCLES = 20_000_000          # ~10 ms


def concurrent_stream(device, priority: Optional[int] = None, tries: int = 8) -> "torch.cuda.Stream":
    """A new stream on ``device`` that executes concurrently with the CURRENT stream (verified), or the last candidate tried."""
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    main = torch.cuda.current_stream(dev)
    cand = None
    for _ in range(max(1, tries)):
        cand = torch.cuda.Stream(device=dev) if priority is None else torch.cuda.Stream(device=dev, priority=int(priority))
        try:
            main.synchronize()
            done_main, done_cand = torch.cuda.Event(), torch.cuda.Event()
            torch.cuda._sleep(_SPIN_CYCLES)
            done_main.record(main)
            with torch.cuda.stream(cand):
                x = torch.zeros(8, device=dev)
                x.add_(1)
                done_cand.record(cand)
            done_cand.synchronize()
            overlapped = not done_main.query()
            main.synchronize()
        except Exception:               # no spin kernel on this build: take the stream as it is
            return cand
        if overlapped:
            return cand
    return cand
