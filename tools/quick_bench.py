#!/usr/bin/env python3
"""Ad-hoc timing of the on-device decode loop (development aid; bench.py is the contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))
import torch
from fq3hip.config import qwen3_tts_0p6b, qwen3_tts_1p7b
from fq3hip.weights import synth_weights, synth_prompt
from fq3hip.engine import Fq3Engine

def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "0.6b"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    cfg = qwen3_tts_0p6b() if size == "0.6b" else qwen3_tts_1p7b()
    dt = torch.bfloat16
    t0 = time.time()
    W = synth_weights(cfg, 0, dt, parts=("talker", "predictor"))
    print(f"weights synth {time.time()-t0:.1f}s", flush=True)
    eng = Fq3Engine(cfg, W, "cuda", dt, max_seq_len=2048, max_frames=2048)
    tie, tam, tth, tpe, _ = synth_prompt(cfg, 200, 32, 0, dtype=dt)
    x = tie[0].cuda().contiguous()
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.time(); logits, hidden = eng.prefill(x); torch.cuda.synchronize()
        print(f"prefill(200 tokens, MFMA) {1e3*(time.time()-t0):.1f} ms")
    V = cfg.talker.vocab_size
    kw = dict(temperature=0.9, top_k=50, top_p=1.0, do_sample=True)
    nf = 64
    tn = torch.empty(nf, V, dtype=dt, device="cuda").exponential_(1)
    pn = torch.empty(nf, 15, cfg.predictor.vocab_size, dtype=dt, device="cuda").exponential_(1)
    fn = torch.empty(V, dtype=dt, device="cuda").exponential_(1)
    for graph in (False, True):
        tok = eng.sample(logits, sup_lo=V - 1024, sup_hi=V, keep_id=cfg.codec_eos_token_id, suppress_eos=True, noise=fn, **kw)
        eng.decode_begin(first_token=int(tok), prefill_len=200, gen_step=0, past_hidden=hidden,
                         trailing_text=tth[0].cuda().contiguous(), tts_pad_embed=tpe.view(-1).cuda().contiguous(),
                         repetition_penalty=1.05, min_new_tokens=frames, max_new_tokens=frames,
                         talker_noise=tn, pred_noise=pn, noise_frames=nf, **kw)
        if graph:
            t0 = time.time(); eng.graph_capture(); print(f"graph capture {1e3*(time.time()-t0):.1f} ms")
        else:
            eng.graph_reset()
        eng.decode_frames(8); n, d = eng.decode_poll()
        t0 = time.time()
        left = frames - 8
        while left > 0:
            k = min(8, left); eng.decode_frames(k); left -= k
            n, d = eng.decode_poll()
        el = time.time() - t0
        print(f"graph={graph}: {n} frames, {1e3*el/(frames-8):.3f} ms/frame, RTF(decode only) {(frames-8)*0.08/el:.1f}, done={d}")
    codes = eng.decode_codes(0, n)
    print(codes[:2].tolist())

if __name__ == "__main__":
    main()
