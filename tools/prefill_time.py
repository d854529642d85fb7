#!/usr/bin/env python3
"""Times the short-prompt (200-token) prefill with the weight-stationary GEMMs on / off (`fq3_set_option("skinny_gemm", v)`),
compares the two first-token logits / hidden rows, and -- with `trace` -- just runs a few prefills for a `rocprofv3 --kernel-trace`.
usage: prefill_time.py [0p6b|1p7b] [trace]      (development aid; bench.py is the contract)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "0p6b"
    dev = "cuda:0"
    cfg, model = bench.build_model(dev, size)
    req = bench.build_request(cfg, dev)
    x = bench.prepared_prompt(model, req)[0][0].contiguous()
    eng = model.talker_graph.engine
    if "trace" in sys.argv:
        for v in (0, 1):
            eng.set_option("skinny_gemm", v)
            for _ in range(3):
                eng.prefill(x)
        torch.cuda.synchronize()
        return
    outs = {}
    for v in (0, 1, 0, 1):
        eng.set_option("skinny_gemm", v)
        for _ in range(3):
            lg, hd = eng.prefill(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lg, hd = eng.prefill(x)
        e1.record(); torch.cuda.synchronize()
        outs[v] = (lg.float().clone(), hd.float().clone())
        print(f"{size} prefill of {x.shape[0]} tokens, skinny_gemm={v}: {e0.elapsed_time(e1) / 20:.3f} ms")
    (l0, h0), (l1, h1) = outs[0], outs[1]
    print(f"first-token logits: max |d| {float((l0 - l1).abs().max()):.4f} of scale {float(l0.abs().max()):.3f}; argmax {int(l0.argmax())} vs {int(l1.argmax())}; "
          f"hidden max |d| {float((h0 - h1).abs().max()):.4f} of {float(h0.abs().max()):.3f}")


if __name__ == "__main__":
    main()
