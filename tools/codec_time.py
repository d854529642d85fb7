#!/usr/bin/env python3
"""Codec decoder timing (development aid): full decode of 370 frames (two pieces: 300 + 95 with context), one 300-frame piece,
a streaming phase-2 chunk (25 context + 8 new frames, tail decode) and the first streaming chunk (178 frames in, 8 out); then the
BATCHED forms (fq3_codec_decode_batch): B utterances of 370 frames / B first chunks through one launch set, per utterance.
usage: codec_time.py [bf16|bf16x2|fp32] [batch sizes, e.g. 4,16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))
import torch
from fq3hip.config import qwen3_tts_0p6b
from fq3hip.weights import synth_weights
from fq3hip.codec import HipSpeechTokenizer


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    Bs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,16").split(",")]
    cap8 = int(sys.argv[3]) if len(sys.argv) > 3 else 0             # measurement hook: fq3_codec_set_option("glds_cap8")
    cfg = qwen3_tts_0p6b()
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("codec",), codec_normalized=True)
    tok = HipSpeechTokenizer(cfg.codec, W, "cuda", max_frames=400, precision=prec)
    if cap8:
        tok.set_option("glds_cap8", cap8)
        print(f"glds_cap8 = {cap8}")
    g = torch.Generator().manual_seed(4)
    codes = torch.randint(0, cfg.codec.codebook_size, (370, 16), generator=g).cuda()
    n33 = tok.num_samples_total(33); n178 = tok.num_samples_total(178)
    for fuse in ((0, 1, 2) if prec == "bf16" else ((0, 1) if prec == "bf16x2" else (1,))):
      tok.set_option("fuse_units", fuse)
      print(f"fused residual units = {fuse}: ", end="")
      print(f"precision {prec}: full 370 frames {timed(lambda: tok.decode_tensor(codes)):.3f} ms | one piece of 300 frames {timed(lambda: tok.decode_tensor(codes[:300].contiguous())):.3f} ms | "
          f"chunk 25+8 frames (tail of 8) {timed(lambda: tok.decode_tensor(codes[:33].contiguous(), n33 - 8 * 1920), 10):.3f} ms | "
          f"first chunk 170+8 frames (tail of 8) {timed(lambda: tok.decode_tensor(codes[:178].contiguous(), n178 - 8 * 1920), 10):.3f} ms", flush=True)
    tok.set_option("fuse_units", 1)
    cut = int(170 / 370 * tok.num_samples_total(370))
    # the same decodes behind the reference's cached front-end state (fq3_codec_prefix_*, round 6): bit-identical waveforms
    pf = tok.prefix_for(codes[:170].contiguous())
    print(f"precision {prec} with the prefix state of the 170 reference frames: first chunk 170+8 (tail of 8) "
          f"{timed(lambda: tok.decode_tensor(codes[:178].contiguous(), n178 - 8 * 1920, prefix=pf), 10):.3f} ms | 170+32 (tail of 8) "
          f"{timed(lambda: tok.decode_tensor(codes[:202].contiguous(), tok.num_samples_total(202) - 8 * 1920, prefix=pf), 10):.3f} ms "
          f"(without: {timed(lambda: tok.decode_tensor(codes[:202].contiguous(), tok.num_samples_total(202) - 8 * 1920), 10):.3f}) | "
          f"370 frames, tail after the reference {timed(lambda: tok.decode_tensor(codes, cut, prefix=pf)):.3f} ms "
          f"(without: {timed(lambda: tok.decode_tensor(codes, cut)):.3f})", flush=True)
    for B in Bs:
        cb = torch.randint(0, cfg.codec.codebook_size, (B, 370, 16), generator=g).cuda()
        full = timed(lambda: tok.decode_tensor_batch(cb), 3)
        tail = timed(lambda: tok.decode_tensor_batch(cb, cut), 3)
        first = timed(lambda: tok.decode_tensor_batch(cb[:, :178].contiguous(), n178 - 8 * 1920), 5)
        ch = timed(lambda: tok.decode_tensor_batch(cb[:, :33].contiguous(), n33 - 8 * 1920), 5)
        pfs = [tok.prefix_for(cb[b, :170].contiguous()) for b in range(B)]
        tail_p = timed(lambda: tok.decode_tensor_batch(cb, cut, prefixes=pfs), 3)
        first_p = timed(lambda: tok.decode_tensor_batch(cb[:, :178].contiguous(), n178 - 8 * 1920, prefixes=pfs), 5)
        print(f"precision {prec} BATCH of {B} with prefix states: tail after 170 reference frames {tail_p:.3f} ms = {tail_p / B:.3f} per utterance "
              f"(without {tail / B:.3f}) | first chunks 170+8 {first_p:.3f} ms = {first_p / B:.3f} each (without {first / B:.3f})", flush=True)
        print(f"precision {prec} BATCH of {B}: full 370 frames {full:.3f} ms = {full / B:.3f} per utterance | tail after 170 reference frames "
              f"{tail:.3f} ms = {tail / B:.3f} per utterance | first chunks 170+8 (tail of 8) {first:.3f} ms = {first / B:.3f} each | "
              f"chunks 25+8 {ch:.3f} ms = {ch / B:.3f} each", flush=True)


if __name__ == "__main__":
    main()
