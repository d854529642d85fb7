#!/usr/bin/env python3
"""BASELINE config 5 smoke: 1.7B shapes, 4096-token prompt, max_seq_len 6144: prefill time + decode ms/frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))
import torch
from fq3hip.config import qwen3_tts_1p7b
from fq3hip.weights import synth_weights, synth_prompt
from fq3hip.engine import Fq3Engine
from fq3hip.generate import run_frames

cfg = qwen3_tts_1p7b()
dt = torch.bfloat16
W = synth_weights(cfg, 0, dt, parts=("talker", "predictor"))
eng = Fq3Engine(cfg, W, "cuda", dt, max_seq_len=6144, max_frames=256)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tie, tam, tth, tpe, _ = synth_prompt(cfg, L, 32, 0, dtype=dt)
x = tie[0].cuda().contiguous()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time(); logits, hidden = eng.prefill(x); torch.cuda.synchronize()
    print(f"prefill({L} tokens, 1.7B) {1e3*(time.time()-t0):.1f} ms   finite={bool(torch.isfinite(logits.float()).all())}")
V = cfg.talker.vocab_size
tok = eng.sample(logits, temperature=1.0, top_k=0, top_p=1.0, do_sample=False, sup_lo=V - 1024, sup_hi=V, keep_id=cfg.codec_eos_token_id, suppress_eos=True)
eng.set_predictor_sampling(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
eng.decode_begin(first_token=int(tok), prefill_len=L, gen_step=0, past_hidden=hidden, trailing_text=tth[0].cuda().contiguous(),
                 tts_pad_embed=tpe.view(-1).cuda().contiguous(), temperature=1.0, top_k=0, top_p=1.0, do_sample=False,
                 repetition_penalty=1.0, min_new_tokens=128, max_new_tokens=128)
eng.graph_capture()
eng.decode_frames(16); eng.decode_poll()
t0 = time.time(); eng.decode_frames(64); n, d = eng.decode_poll(); el = time.time() - t0
print(f"decode at KV ~{L+48}: {1e3*el/64:.3f} ms/frame ({n} frames, done={d})")
