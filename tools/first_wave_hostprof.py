#!/usr/bin/env python3
"""cProfile of the host thread during the first wave of N simultaneous streaming requests (development aid): where do the ~100 ms of
host time in front of the first frame go?   usage: first_wave_hostprof.py [lanes=128]"""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def main():
    lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = "cuda:0"
    cfg, model = bench.build_model(dev, codec_precision=bench.HEADLINE_CODEC)
    req = bench.build_request(cfg, dev)

    def one(n_utt, stop_after_first=True):
        torch.manual_seed(4242)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seen = set()
        t_first = None
        for i, audio, sr, tm in model.generate_voice_clone_batch_streaming(
                [req["text"]] * n_utt, language=req["language"], ref_text=req["ref_text"], voice_clone_prompt=req["voice_clone_prompt"],
                instruct=req["instruct"], chunk_size=bench.CHUNK, max_new_tokens=24, min_new_tokens=24, lanes=lanes):
            if i not in seen:
                seen.add(i)
                if t_first is None:
                    t_first = 1e3 * (time.perf_counter() - t0)
                if len(seen) == n_utt:
                    t_last = 1e3 * (time.perf_counter() - t0)
        return t_first, t_last

    one(lanes); one(lanes)
    pr = cProfile.Profile()
    pr.enable()
    r = one(lanes)
    pr.disable()
    print(f"first chunk of the first / last request: {r[0]:.1f} / {r[1]:.1f} ms ({lanes} requests, 24 frames each, profiled)")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
    print(s.getvalue()[:7000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats("batching|generate|engine|talker_graph|prompt|native_model|model.py", 60)
    print(s.getvalue()[:12000])


if __name__ == "__main__":
    main()
