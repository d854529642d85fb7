#!/usr/bin/env python3
"""Golden PCM of the CPU oracle's 12 Hz codec decoder at the REAL shapes bench.py runs (latent 1024, decoder_dim 1536,
8 transformer layers, head_dim 64, sliding window 72, upsampling 2*2*8*5*4*3 = 1920), on the *normalised* synthetic
weights (fq3hip.weights.synth_weights(codec_normalized=True): activations stay O(1) like a trained vocoder).

    python oracle/make_golden_codec_real.py        # -> tests/golden/codec_real.npz (~1.5 MB)

Cases: T = 40 (inside the attention window) and T = 100 (> window 72: the oldest keys fall out).  Stored: codes, the
fp32 oracle PCM, and the bf16 oracle PCM as raw bf16 bits (exact).  A second file, tests/golden/codec_real_q.npz, holds the
fp32-ARITHMETIC oracle PCM on the bf16-VALUED weights (what a bf16 checkpoint evaluated without activation rounding gives):
the reference point of the high-precision codec mode (fq3hip codec_precision="fp32": bf16 checkpoint weights, fp32
activations and products), which separates the weight-quantisation share of the bf16 error from the arithmetic share.
Test infrastructure only.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

from fq3hip.config import qwen3_tts_0p6b            # noqa: E402
from fq3hip.weights import synth_weights            # noqa: E402
from oracle import qwen3tts_oracle as O             # noqa: E402

CASES = (40, 100)


def main():
    cfg = qwen3_tts_0p6b()
    c = cfg.codec
    W32 = synth_weights(cfg, 0, torch.float32, parts=("codec",), codec_normalized=True)
    Wb = {k: v.to(torch.bfloat16) for k, v in W32.items()}
    Wq = {k: v.float() for k, v in Wb.items()}           # bf16-valued weights, fp32 arithmetic
    out, outq = {}, {}
    for T in CASES:
        g = torch.Generator().manual_seed(100 + T)
        codes = torch.randint(0, c.codebook_size, (T, c.num_quantizers), generator=g)
        with torch.inference_mode():
            p32 = O.codec_decode(codes, W32, c)
            pb = O.codec_decode(codes, Wb, c)
            pq = O.codec_decode(codes, Wq, c)
        outq[f"codes_{T}"] = codes.numpy().astype(np.int16)
        outq[f"pcm_f32q_{T}"] = pq.numpy().astype(np.float32)
        print(f"T={T}: fp32 arithmetic on bf16-valued weights vs fp32 weights RMS {float((pq - p32).pow(2).mean().sqrt()):.3e}, "
              f"bf16 arithmetic vs fp32 arithmetic on the same bf16-valued weights RMS {float((pb.float() - pq).pow(2).mean().sqrt()):.3e}")
        out[f"codes_{T}"] = codes.numpy().astype(np.int16)
        out[f"pcm_f32_{T}"] = p32.numpy().astype(np.float32)
        out[f"pcm_bf16bits_{T}"] = pb.view(torch.int16).numpy()
        d = float((pb.float() - p32).pow(2).mean().sqrt())
        print(f"T={T}: {p32.numel()} samples, pcm std {float(p32.std()):.4f}, oracle bf16 vs fp32 RMS {d:.3e}")
    path = os.path.join(ROOT, "tests", "golden", "codec_real.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))
    pathq = os.path.join(ROOT, "tests", "golden", "codec_real_q.npz")
    np.savez_compressed(pathq, **outq)
    print("wrote", pathq, os.path.getsize(pathq))


if __name__ == "__main__":
    main()
