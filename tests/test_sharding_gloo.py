"""CPU, world_size 2, gloo: the utterance-sharding + result-gather path used at N > 1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fq3hip.sharding import shard_indices, gather_arrays, run_sharded


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ragged = gather_arrays(np.arange(3 + 2 * rank, dtype=np.float32) + 100 * rank)
        items = list(range(5))
        out = run_sharded(items, lambda i: np.full(i + 1, float(i), dtype=np.float32))
        q.put((rank, [a.tolist() for a in ragged], [a.tolist() for a in out]))
    finally:
        dist.destroy_process_group()


def test_shard_indices_round_robin():
    assert shard_indices(64, 3, 8) == list(range(3, 64, 8))
    assert sorted(sum((shard_indices(10, r, 4) for r in range(4)), [])) == list(range(10))


def test_gather_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ragged, out in res:
        assert ragged == [[0.0, 1.0, 2.0], [100.0, 101.0, 102.0, 103.0, 104.0]]
        assert out == [[float(i)] * (i + 1) for i in range(5)]


def test_single_process_passthrough():
    assert [a.tolist() for a in gather_arrays(np.array([1.0, 2.0]))] == [[1.0, 2.0]]
    out = run_sharded([3, 4], lambda v: np.array([v], dtype=np.float32))
    assert [a.tolist() for a in out] == [[3.0], [4.0]]
