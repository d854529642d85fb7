#!/usr/bin/env python3
"""Times the 200-token prefill with the resident-tile attention kernel on / off (`fq3_set_option("flash_small", v)`; bit-identical
outputs) and a PACKED prefill of 10 such prompts over one KV pool (the staged admission of the batch scheduler: `fq3_prefill_batch`).
usage: prefill_small_time.py [0p6b|1p7b] [prompts per packed prefill = 10]      (development aid; bench.py is the contract)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from fq3hip.engine import Fq3Engine, Fq3KvPool


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "0p6b"
    dev = "cuda:0"
    cfg, model = bench.build_model(dev, size)
    req = bench.build_request(cfg, dev)
    x = bench.prepared_prompt(model, req)[0][0].contiguous()
    eng = model.talker_graph.engine
    outs = {}
    for v in (0, 1, 0, 1):
        eng.set_option("flash_small", v)
        ms = timed(lambda: eng.prefill(x))
        lg, hd = eng.prefill(x)
        outs[v] = (lg.float().clone(), hd.float().clone())
        print(f"{size} prefill of {x.shape[0]} tokens, flash_small={v}: {ms:.3f} ms", flush=True)
    print("identical logits / hidden:", bool(torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10          # prompts per packed prefill (the scheduler's first wave stages slices of 16)
    pool = Fq3KvPool(eng.cfg, n * 5, device=dev, dtype=eng.dtype)
    # (the packed rows of all prompts go through the FIRST context's prefill workspaces: sized by its max_seq_len)
    msl = max(int(eng.max_seq_len), n * int(x.shape[0]) + 8)
    engs = [Fq3Engine(eng.cfg, eng.weights, device=dev, dtype=eng.dtype, max_seq_len=msl, max_frames=8, share=eng, pool=pool) for _ in range(n)]
    engs[0].prefill_reserve()
    xs = [x] * n
    for v in (0, 1, 0, 1):
        for e in engs:
            e.set_option("flash_small", v)
        ms = timed(lambda: Fq3Engine.prefill_batch(engs, xs), 10)
        print(f"{size} PACKED prefill of {n} x {x.shape[0]} tokens (one pool), flash_small={v}: {ms:.3f} ms = {ms / n:.3f} per prompt", flush=True)


if __name__ == "__main__":
    main()
