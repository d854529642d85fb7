#!/usr/bin/env python3
"""First-chunk latency of SIMULTANEOUS streaming requests through the lock-step batch (development aid; bench.py is the contract).
`n` requests are handed to generate_voice_clone_batch_streaming at once; for every request the time from the call to its first audio
chunk on the host is recorded.  Sweeps the scheduler's first wave (BatchDecoder.first_wave: requests prepared + prefilled + armed before
the first frame is queued).
usage: batch_ttfa_probe.py [lanes list, e.g. 32,64,128] [first_wave list, e.g. 0,16,32,64 (0 = one per lane)] [utterances per lane = 2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench


def run(model, req, lanes, n_utt, first_wave):
    model.batch_first_wave_streaming = first_wave
    torch.manual_seed(4242)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first, samples = {}, 0
    for i, audio, sr, tm in model.generate_voice_clone_batch_streaming(
            [req["text"]] * n_utt, language=req["language"], ref_text=req["ref_text"], voice_clone_prompt=req["voice_clone_prompt"],
            instruct=req["instruct"], chunk_size=bench.CHUNK, max_new_tokens=bench.FRAMES, min_new_tokens=bench.FRAMES, lanes=lanes):
        first.setdefault(i, time.perf_counter() - t0)
        samples += len(audio)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    w = np.sort(np.asarray([first[i] for i in range(min(lanes, n_utt))])) * 1e3
    allr = np.sort(np.asarray(list(first.values()))) * 1e3
    q = lambda a, p: float(a[min(len(a) - 1, int(p * len(a)))])
    return dict(lanes=lanes, first_wave=first_wave, utterances=n_utt, rtf=round(samples / 24000.0 / wall, 1),
                first_lanes_ms=dict(p25=round(q(w, 0.25), 1), p50=round(q(w, 0.5), 1), p75=round(q(w, 0.75), 1), max=round(float(w[-1]), 1),
                                    under_150=int((w < 150).sum())),
                all_ms=dict(p50=round(q(allr, 0.5), 1), max=round(float(allr[-1]), 1)))


def main():
    lanes_l = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32,64,128").split(",")]
    fw_l = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,32").split(",")]
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dev = "cuda:0"
    cfg, model = bench.build_model(dev, codec_precision=bench.HEADLINE_CODEC)
    req = bench.build_request(cfg, dev)
    for lanes in lanes_l:
        run(model, req, lanes, lanes, None)                       # warm-up: contexts, graph capture, workspaces
        for fw in fw_l:
            r = run(model, req, lanes, per * lanes, fw if fw > 0 else None)
            print(r, flush=True)


if __name__ == "__main__":
    main()
