#!/usr/bin/env python3
"""Workload for a kernel trace of ONE shape of the codec decoder (development aid): the streaming phase-2 chunk (25 context + 8 new
frames, tail decode) by default, or the 200-frame tail of a batch behind prefix states.
usage: codec_chunk_trace.py [bf16|bf16x2] [chunk|first|first32|tail16] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))
import torch
from fq3hip.config import qwen3_tts_0p6b
from fq3hip.weights import synth_weights
from fq3hip.codec import HipSpeechTokenizer


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x2"
    what = sys.argv[2] if len(sys.argv) > 2 else "chunk"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    cfg = qwen3_tts_0p6b()
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("codec",), codec_normalized=True)
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", max_frames=400, precision=prec)
    g = torch.Generator().manual_seed(4)
    codes = torch.randint(0, cfg.codec.codebook_size, (370, 16), generator=g).cuda()
    if what == "first32":
        B = 32
        cb = codes[:178].unsqueeze(0).repeat(B, 1, 1).contiguous()
        pf = tok.prefix_for(codes[:170].contiguous())
        first = tok.num_samples_total(178) - 8 * 1920
        fn = lambda: tok.decode_tensor_batch(cb, first, prefixes=[pf] * B)
    elif what == "tail16":
        B = 16
        cb = codes.unsqueeze(0).repeat(B, 1, 1).contiguous()
        first = tok.num_samples_total(370) - 200 * 1920
        fn = lambda: tok.decode_tensor_batch(cb, first)
    else:
        n = 33 if what == "chunk" else 178
        c = codes[:n].contiguous()
        first = tok.num_samples_total(n) - 8 * 1920
        fn = lambda: tok.decode_tensor(c, first)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{prec} {what}: {e0.elapsed_time(e1) / reps:.3f} ms per call over {reps} calls")


if __name__ == "__main__":
    main()
