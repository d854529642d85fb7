#!/usr/bin/env python3
"""Batched decode timing (development aid): B lanes of the 0.6B-shape model in lock-step, one hipGraph per frame.
usage: batch_bench.py [B=8] [frames=48]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))
import torch
from fq3hip.config import qwen3_tts_0p6b
from fq3hip.weights import synth_weights, synth_prompt
from fq3hip.engine import Fq3Engine, Fq3Batch


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    cfg = qwen3_tts_0p6b()
    dt = torch.bfloat16
    W = synth_weights(cfg, 0, dt, parts=("talker", "predictor"))
    first = Fq3Engine(cfg, W, "cuda", dt, max_seq_len=1024, max_frames=256)
    lanes = [first] + [Fq3Engine(cfg, W, "cuda", dt, max_seq_len=1024, max_frames=256, share=first) for _ in range(B - 1)]
    V, Vp = cfg.talker.vocab_size, cfg.predictor.vocab_size
    kw = dict(temperature=0.9, top_k=50, top_p=1.0, do_sample=True)
    nf = 64
    keep = []
    for i, eng in enumerate(lanes):
        tie, tam, tth, tpe, _ = synth_prompt(cfg, 200, 32, 0, dtype=dt, seed=1234 + 10 * i)
        logits, hidden = eng.prefill(tie[0].cuda().contiguous())
        tn = torch.empty(nf, V, dtype=dt, device="cuda").exponential_(1)
        pn = torch.empty(nf, 15, Vp, dtype=dt, device="cuda").exponential_(1)
        fn = torch.empty(V, dtype=dt, device="cuda").exponential_(1)
        tok = eng.sample(logits, sup_lo=V - 1024, sup_hi=V, keep_id=cfg.codec_eos_token_id, suppress_eos=True, noise=fn, **kw)
        eng.decode_begin(first_token=int(tok), prefill_len=200, gen_step=0, past_hidden=hidden,
                         trailing_text=tth[0].cuda().contiguous(), tts_pad_embed=tpe.view(-1).cuda().contiguous(),
                         repetition_penalty=1.05, min_new_tokens=250, max_new_tokens=250,
                         talker_noise=tn, pred_noise=pn, noise_frames=nf, **kw)
        keep.append((tn, pn))
    batch = Fq3Batch(lanes)
    batch.graph_capture()
    batch.frames(8); torch.cuda.synchronize()
    t0 = time.perf_counter(); batch.frames(frames); torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / frames
    n = [e.decode_poll()[0] for e in lanes]
    print(f"B={B}: {ms:.3f} ms per lock-step frame -> {B * 80.0 / ms:.1f}x real-time aggregate ({80.0 / ms:.1f}x per lane), frames per lane {n}")
    print("lane 0 first frames:", lanes[0].decode_codes(0, 2).tolist())


if __name__ == "__main__":
    main()
