"""Utterance sharding over the GPUs of one node.

The path shards by independent units: every rank holds a full replica of the weights and its own
decode context, utterance i goes to rank ``i % world``, and nothing of one utterance is shared with
another.  The ONLY collectives are the result gather (lengths ``all_gather`` then padded payload
``all_gather``) -- RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.  The reference has
no multi-GPU code at all (requests are serialised by a lock, examples/openai_server.py:71).
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment: item i -> rank i % world."""
    return list(range(rank, n_items, world))


def gather_arrays(local: np.ndarray, device="cpu", always_collective: bool = False) -> List[np.ndarray]:
    """All ranks contribute one 1-D array (ragged); every rank gets the list ordered by rank.  A single rank needs no
    collective and gets its array back, unless ``always_collective`` asks for the real ``all_gather`` pair anyway (how the
    1-GPU test makes RCCL itself execute)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not always_collective):
        return [np.asarray(local)]
    world = dist.get_world_size()
    t = torch.as_tensor(np.ascontiguousarray(local)).to(device)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=device)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(x) for x in lens]
    mx = max(max(lens), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=device)
    pad[: t.numel()] = t.reshape(-1)
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [b[:l].cpu().numpy() for b, l in zip(bufs, lens)]


def run_sharded(items: Sequence, fn: Callable, device="cpu"):
    """Run ``fn(item) -> 1-D np.ndarray`` on this rank's shard and gather every result to every rank,
    returned in the original item order.  ONE result gather at the end (a lengths header travels in the same payload):
    the decode itself never waits on another rank."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = shard_indices(len(items), rank, world)
    results = [np.asarray(fn(items[i]), dtype=np.float32).reshape(-1) for i in mine]
    header = np.asarray([len(r) for r in results], dtype=np.float32)        # item lengths < 2^24: exact in fp32
    payload = np.concatenate([header] + results) if results else header
    out: List = [None] * len(items)
    for rk, flat in enumerate(gather_arrays(payload, device)):
        idx = shard_indices(len(items), rk, world)
        lens = [int(x) for x in flat[:len(idx)]]
        pos = len(idx)
        for i, n in zip(idx, lens):
            out[i] = flat[pos:pos + n]
            pos += n
    return out
