#!/usr/bin/env python3
"""Where the time-to-first-audio goes: the stages of one streaming voice-clone request, each closed by a device sync
(so the sum slightly exceeds the real, pipelined TTFA that bench.py reports)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))
import numpy as np, torch
import bench
from fq3hip.generate import _prefill_and_arm, run_frames

dev = torch.device("cuda:0")
cfg, model = bench.build_model(dev)
req = bench.build_request(cfg, dev)
for i in range(3):
    bench.one_utterance(model, req, 10 + i, frames=24)
sync = torch.cuda.synchronize
m = model.model.model
tok = m.speech_tokenizer
rows = []
for rep in range(5):
    torch.manual_seed(100 + rep)
    sync(); t0 = time.perf_counter()
    _m, talker, config, tie, tam, tth, tpe, rc = model._prepare_generation(
        text=req["text"], language=req["language"], ref_text=req["ref_text"], voice_clone_prompt=req["voice_clone_prompt"],
        instruct=req["instruct"], non_streaming_mode=False)
    sync(); t1 = time.perf_counter()
    eng, tn, pn, mf = _prefill_and_arm(talker, tie, tam, tth, tpe, config, model.predictor_graph, model.talker_graph,
                                       200, 200, 0.9, 50, 1.0, True, 1.05, use_graph=True)
    sync(); t2 = time.perf_counter()
    run_frames(eng, tn, pn, 0, 8)
    n, done = eng.decode_poll()
    t3 = time.perf_counter()
    codes = eng.decode_codes(0, 8)
    full = torch.cat([rc, codes], 0)
    cut = int(rc.shape[0] / full.shape[0] * tok.num_samples_total(full.shape[0]))
    pcm = tok.decode_tensor(full, cut).cpu().numpy()
    t4 = time.perf_counter()
    rows.append([1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3)])
r = np.median(np.array(rows), axis=0)
print(f"tokenise + prompt build {r[0]:.2f} ms | prefill + first token + arm {r[1]:.2f} ms | 8 frames + poll {r[2]:.2f} ms | "
      f"first chunk vocoder ({int(full.shape[0])} frames in, tail out) + D2H {r[3]:.2f} ms | sum {r.sum():.2f} ms")
tt = [bench.one_utterance(model, req, 200 + i, frames=24)[0] * 1e3 for i in range(8)]
print(f"public generate_voice_clone_streaming TTFA p50 {np.median(tt):.2f} ms")
