#!/usr/bin/env python3
"""Golden vectors for the long-prompt configuration (BASELINE configs[4]: 4k-token prompt, 1.7B shapes): the CPU oracle's
prefill of a 4096-token prompt through 2 talker layers at the real 1.7B layer dims (hidden 2048, 16:8 heads of 128,
intermediate 6144), last-position hidden state and logits, plus two decode steps on top of the 4096-key cache.

    python oracle/make_golden_longprompt.py          # -> tests/golden/longprompt.npz

Test infrastructure only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

from fq3hip.config import qwen3_tts_1p7b               # noqa: E402
from fq3hip.weights import synth_weights, synth_prompt  # noqa: E402
from oracle import qwen3tts_oracle as O                 # noqa: E402

L = 4096


def config():
    cfg = qwen3_tts_1p7b()
    cfg.talker.num_hidden_layers = 2
    cfg.predictor.num_hidden_layers = 1
    return cfg


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = config()
    out = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        W = synth_weights(cfg, 0, dtype, parts=("talker",))
        tie, tam, _, _, _ = synth_prompt(cfg, L, 4, 0, dtype=dtype)
        tie = (tie * 30).to(dtype)                       # O(1) activations
        orc = O.OracleTTS(cfg, W, max_seq_len=L + 8)
        with torch.inference_mode():
            logits, hidden, _, n = orc.prefill(tie, tam)
            out[f"logits_{tag}"] = logits.float().numpy()
            out[f"hidden_{tag}"] = hidden.float().view(-1).numpy()
            g = torch.Generator().manual_seed(3)
            for step in range(2):
                x = torch.randn(1, 1, cfg.talker.hidden_size, generator=g).to(dtype)
                out[f"step{step}_{tag}"] = orc.talker_step(x, L + step).float().view(-1).numpy()
        print(tag, "hidden max", float(hidden.float().abs().max()), "logits max", float(logits.float().abs().max()))
    path = os.path.join(ROOT, "tests", "golden", "longprompt.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
