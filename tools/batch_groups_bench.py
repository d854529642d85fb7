#!/usr/bin/env python3
"""Do two lock-step batches overlap on one GPU?  G groups x B lanes, each group on its own HIP stream, frames enqueued
back to back from one host thread (no Python between the launches), one sync at the end.  usage: batch_groups_bench.py [G=2] [B=8]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))
import torch
from fq3hip.config import qwen3_tts_0p6b
from fq3hip.weights import synth_weights, synth_prompt
from fq3hip.engine import Fq3Engine, Fq3Batch


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    frames = 48
    cfg = qwen3_tts_0p6b()
    dt = torch.bfloat16
    W = synth_weights(cfg, 0, dt, parts=("talker", "predictor"))
    first = Fq3Engine(cfg, W, "cuda", dt, max_seq_len=1024, max_frames=256)
    V, Vp = cfg.talker.vocab_size, cfg.predictor.vocab_size
    kw = dict(temperature=0.9, top_k=50, top_p=1.0, do_sample=True)
    groups, keep = [], []
    for g in range(G):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            lanes = [first if (g == 0 and i == 0) else Fq3Engine(cfg, W, "cuda", dt, max_seq_len=1024, max_frames=256, share=first) for i in range(B)]
            for i, eng in enumerate(lanes):
                tie, tam, tth, tpe, _ = synth_prompt(cfg, 200, 32, 0, dtype=dt, seed=1234 + 10 * i + 100 * g)
                logits, hidden = eng.prefill(tie[0].cuda().contiguous())
                tn = torch.empty(64, V, dtype=dt, device="cuda").exponential_(1)
                pn = torch.empty(64, 15, Vp, dtype=dt, device="cuda").exponential_(1)
                fn = torch.empty(V, dtype=dt, device="cuda").exponential_(1)
                tok = eng.sample(logits, sup_lo=V - 1024, sup_hi=V, keep_id=cfg.codec_eos_token_id, suppress_eos=True, noise=fn, **kw)
                eng.decode_begin(first_token=int(tok), prefill_len=200, gen_step=0, past_hidden=hidden,
                                 trailing_text=tth[0].cuda().contiguous(), tts_pad_embed=tpe.view(-1).cuda().contiguous(),
                                 repetition_penalty=1.05, min_new_tokens=250, max_new_tokens=250, talker_noise=tn, pred_noise=pn,
                                 noise_frames=64, **kw)
                keep.append((tn, pn, tth, tpe, hidden))
            batch = Fq3Batch(lanes)
            batch.graph_capture()
            batch.frames(4)
            st.synchronize()
        groups.append((st, batch, lanes))
    torch.cuda.synchronize()
    for active in range(1, G + 1):
        t0 = time.perf_counter()
        for st, batch, _ in groups[:active]:
            with torch.cuda.stream(st):
                batch.frames(frames if active == G else 8)
        torch.cuda.synchronize()
        dtm = 1e3 * (time.perf_counter() - t0) / (frames if active == G else 8)
        print(f"{active} group(s) x {B} lanes in flight: {dtm:.3f} ms per lock-step frame round -> {active * B * 80.0 / dtm:.1f}x real-time aggregate")


if __name__ == "__main__":
    main()
