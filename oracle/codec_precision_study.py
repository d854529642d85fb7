#!/usr/bin/env python3
"""Which stages of the 12 Hz codec decoder need more than bf16 activations?  (Test infrastructure / analysis only -- see the header
of oracle/qwen3tts_oracle.py; nothing in the product imports this.)

The decoder is evaluated on the CPU with the bf16-VALUED checkpoint weights and fp32 arithmetic (the reference point of
tests/golden/codec_real_q.npz), and again with the activations of a chosen SET OF STAGES rounded to bf16 after every layer of those
stages (what a bf16 kernel stores), everything else left in fp32 (what the bf16x2 kernels keep to ~1e-5).  The PCM RMS difference to
the all-fp32 run says how much of the 7.6e-3 of the all-bf16 decode each stage is responsible for, i.e. which stages a MIXED
vocoder mode could run on the cheaper bf16 kernels while staying inside the 1e-3 bound.

    python oracle/codec_precision_study.py [T=100]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

from fq3hip.config import qwen3_tts_0p6b            # noqa: E402
from fq3hip.weights import synth_weights            # noqa: E402
from oracle import qwen3tts_oracle as O             # noqa: E402

STAGES = ("front", "transformer", "upsample", "dec0", "block1", "block2", "block3", "block4", "final")


def decode(codes, W, c, low):
    """codec_decode of oracle/qwen3tts_oracle.py:509 with the activations of the stages in `low` rounded to bf16 after every layer."""
    r = lambda st, x: x.to(torch.bfloat16).float() if st in low else x
    p = "decoder"
    h = r("front", O.rvq_decode(codes, W, c))
    h = r("front", O.causal_conv1d(h, W[f"{p}.pre_conv.conv.weight"], W[f"{p}.pre_conv.conv.bias"]))
    # transformer: round the residual stream after every sub-layer (what the bf16 kernels store between launches)
    t = "decoder.pre_transformer"
    x = h.transpose(0, 1)
    hh = r("transformer", F.linear(x, W[f"{t}.input_proj.weight"], W[f"{t}.input_proj.bias"]))
    T = hh.shape[0]
    d = c.head_dim
    cos, sin = O.rope_cos_sin(torch.arange(T).float(), d, c.rope_theta, hh.dtype)
    qpos, kpos = torch.arange(T)[:, None], torch.arange(T)[None, :]
    mask = (kpos <= qpos) & (kpos > qpos - c.sliding_window)
    for i in range(c.num_hidden_layers):
        q_ = f"{t}.layers.{i}"
        res = hh
        hn = r("transformer", O.rms_norm(hh, W[f"{q_}.input_layernorm.weight"], c.rms_norm_eps))
        q = r("transformer", F.linear(hn, W[f"{q_}.self_attn.q_proj.weight"])).view(T, -1, d)
        k = r("transformer", F.linear(hn, W[f"{q_}.self_attn.k_proj.weight"])).view(T, -1, d)
        v = r("transformer", F.linear(hn, W[f"{q_}.self_attn.v_proj.weight"])).view(T, -1, d)
        q, k = O.apply_rope(q, cos, sin), O.apply_rope(k, cos, sin)
        a = r("transformer", O.attention_fp32(q, k, v, mask, d ** -0.5).reshape(T, -1))
        hh = r("transformer", res + W[f"{q_}.self_attn_layer_scale.scale"] * F.linear(a, W[f"{q_}.self_attn.o_proj.weight"]))
        res = hh
        hn = r("transformer", O.rms_norm(hh, W[f"{q_}.post_attention_layernorm.weight"], c.rms_norm_eps))
        m = r("transformer", F.silu(F.linear(hn, W[f"{q_}.mlp.gate_proj.weight"])) * F.linear(hn, W[f"{q_}.mlp.up_proj.weight"]))
        hh = r("transformer", res + W[f"{q_}.mlp_layer_scale.scale"] * F.linear(m, W[f"{q_}.mlp.down_proj.weight"]))
    hh = r("transformer", O.rms_norm(hh, W[f"{t}.norm.weight"], c.rms_norm_eps))
    h = r("transformer", F.linear(hh, W[f"{t}.output_proj.weight"], W[f"{t}.output_proj.bias"])).transpose(0, 1)
    for i, f in enumerate(c.upsampling_ratios):
        h = r("upsample", O.causal_trans_conv1d(h, W[f"{p}.upsample.{i}.0.conv.weight"], W[f"{p}.upsample.{i}.0.conv.bias"], f))
        h = r("upsample", O.convnext_block(h, W, f"{p}.upsample.{i}.1"))
    dd = f"{p}.decoder"
    h = r("dec0", O.causal_conv1d(h, W[f"{dd}.0.conv.weight"], W[f"{dd}.0.conv.bias"]))
    for i, rate in enumerate(c.upsample_rates):
        st = f"block{i + 1}"
        b = f"{dd}.{i + 1}.block"
        h = r(st, O.snake_beta(h, W[f"{b}.0.alpha"], W[f"{b}.0.beta"]))
        h = r(st, O.causal_trans_conv1d(h, W[f"{b}.1.conv.weight"], W[f"{b}.1.conv.bias"], rate))
        for j, dil in enumerate((1, 3, 9)):
            u = f"{b}.{j + 2}"
            res = h
            y = r(st, O.snake_beta(h, W[f"{u}.act1.alpha"], W[f"{u}.act1.beta"]))
            y = r(st, O.causal_conv1d(y, W[f"{u}.conv1.conv.weight"], W[f"{u}.conv1.conv.bias"], dilation=dil))
            y = r(st, O.snake_beta(y, W[f"{u}.act2.alpha"], W[f"{u}.act2.beta"]))
            y = O.causal_conv1d(y, W[f"{u}.conv2.conv.weight"], W[f"{u}.conv2.conv.bias"])
            h = r(st, y + res)
    n = len(c.upsample_rates)
    h = r("final", O.snake_beta(h, W[f"{dd}.{n + 1}.alpha"], W[f"{dd}.{n + 1}.beta"]))
    h = O.causal_conv1d(h, W[f"{dd}.{n + 2}.conv.weight"], W[f"{dd}.{n + 2}.conv.bias"])
    return h.clamp(min=-1, max=1).reshape(-1)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = qwen3_tts_0p6b()
    c = cfg.codec
    W32 = synth_weights(cfg, 0, torch.float32, parts=("codec",), codec_normalized=True)
    Wq = {k: v.to(torch.bfloat16).float() for k, v in W32.items()}
    g = torch.Generator().manual_seed(100 + T)
    codes = torch.randint(0, c.codebook_size, (T, c.num_quantizers), generator=g)
    rms = lambda a, b: float((a - b).pow(2).mean().sqrt())
    with torch.inference_mode():
        t0 = time.time()
        ref = decode(codes, Wq, c, set())
        chk = O.codec_decode(codes, Wq, c)
        print(f"T={T}: {ref.numel()} samples, {time.time() - t0:.1f} s per two decodes; restated decode vs oracle.codec_decode: {rms(ref, chk):.2e} (must be 0)")
        print(f"all stages bf16: {rms(decode(codes, Wq, c, set(STAGES)), ref):.3e}")
        for st in STAGES:
            print(f"only {st:12s} bf16: {rms(decode(codes, Wq, c, {st}), ref):.3e}", flush=True)
        for low in (("front", "transformer"), ("front", "transformer", "upsample"), ("front", "transformer", "upsample", "dec0"),
                    ("block3", "block4", "final"), ("block4", "final"), ("block2", "block3", "block4", "final")):
            print(f"bf16 in {'+'.join(low):40s}: {rms(decode(codes, Wq, c, set(low)), ref):.3e}", flush=True)


if __name__ == "__main__":
    main()
