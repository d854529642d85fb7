#!/usr/bin/env python3
"""Where the first-chunk latency of N simultaneous streaming requests goes (development aid): host timestamps (ms after the call) of
the scheduler's first milestones -- prompts of the first wave built, their prefills' first tokens on the host, lanes armed, first
frames queued, first poll read, first audio chunk out, last first-chunk out.  usage: batch_ttfa_timeline.py [lanes=128]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import fq3hip.batching as Bt


def main():
    lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = "cuda:0"
    cfg, model = bench.build_model(dev, codec_precision=bench.HEADLINE_CODEC)
    req = bench.build_request(cfg, dev)
    model.batch_first_wave_streaming = None
    marks = []
    t0 = [0.0]
    mark = lambda what: marks.append((what, 1e3 * (time.perf_counter() - t0[0])))

    def wrap(obj, name, label, first_only=True):
        fn = getattr(obj, name)
        seen = [0]

        def w(*a, **k):
            r = fn(*a, **k)
            seen[0] += 1
            if not first_only or seen[0] == 1:
                mark(f"{label} #{seen[0]} done")
            return r
        setattr(obj, name, w)
        return fn

    def one(n_utt):
        torch.manual_seed(4242)
        torch.cuda.synchronize()
        del marks[:]
        t0[0] = time.perf_counter()
        first = {}
        for i, audio, sr, tm in model.generate_voice_clone_batch_streaming(
                [req["text"]] * n_utt, language=req["language"], ref_text=req["ref_text"], voice_clone_prompt=req["voice_clone_prompt"],
                instruct=req["instruct"], chunk_size=bench.CHUNK, max_new_tokens=bench.FRAMES, min_new_tokens=bench.FRAMES, lanes=lanes):
            if i not in first:
                first[i] = 1e3 * (time.perf_counter() - t0[0])
        return first

    one(lanes)                                                     # warm-up
    dec = model._batch_decoder(lanes)
    o1 = wrap(model, "_batch_feed", "prompts of the first wave built (_batch_feed)")
    o2 = wrap(dec, "_stage_many", "staged prefill group (first tokens on the host)", first_only=False)
    o3 = wrap(dec, "_admit", "lane armed (_admit)")
    o4 = wrap(dec.batch, "frames", "frames queued")
    o5 = wrap(dec.batch, "poll_wait", "poll read")
    first = one(lanes)
    w = np.sort(np.asarray(list(first.values())))
    for what, t in marks[:40]:
        print(f"{t:9.2f} ms  {what}")
    print(f"first audio chunks on the host: first {w[0]:.1f} ms, median {w[len(w) // 2]:.1f} ms, last {w[-1]:.1f} ms ({len(w)} requests)")
    print("scheduler thread (s):", {k: round(v, 4) for k, v in dec.host_s.items()})


if __name__ == "__main__":
    main()
