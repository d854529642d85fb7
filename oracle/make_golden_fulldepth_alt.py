#!/usr/bin/env python3
"""A SECOND golden utterance at the benchmarked configurations (full depth), so that the lanes of a lock-step batch are
teacher-forced with different prompts, prompt lengths and positions (tests/test_gpu_batch_fulldepth.py).

    python oracle/make_golden_fulldepth_alt.py        # writes tests/golden/fulldepth_alt.npz (a few minutes on 8 cores)

Same construction as oracle/make_golden_fulldepth.py (seeded synthetic weights at the real shapes, greedy frames, top-1
logit + top-2 margin of every decision) with another prompt: seed 4321, 137 rows (three 64-key tiles, the last one ragged),
16 trailing rows, 16 frames.  Cases: {0.6B, 1.7B} x bf16 and 0.6B fp32.  Test infrastructure only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

from fq3hip.config import qwen3_tts_0p6b, qwen3_tts_1p7b          # noqa: E402
from fq3hip.weights import synth_weights, synth_prompt            # noqa: E402
from oracle import qwen3tts_oracle as O                           # noqa: E402

FRAMES, PROMPT, TRAILING, SEED = 16, 137, 16, 4321


def alt_prompt(cfg, dtype):
    return synth_prompt(cfg, PROMPT, TRAILING, 0, dtype=dtype, seed=SEED)


def run_case(size: str, dtype: torch.dtype):
    cfg = qwen3_tts_0p6b() if size == "0p6b" else qwen3_tts_1p7b()
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    tie, tam, tth, tpe, _ = alt_prompt(cfg, dtype)
    orc = O.OracleTTS(cfg, W, max_seq_len=PROMPT + FRAMES + 8)
    orc.pred_sampling = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    sp = O.SamplingParams(max_new_tokens=FRAMES, **{**O.GREEDY, "min_new_tokens": FRAMES})
    with torch.inference_mode():
        codes = orc.generate(tie, tam, tth, tpe, sp, record_margins=True)
    assert codes.shape == (FRAMES, 16)
    return dict(codes=codes.numpy().astype(np.int32),
                t_margin=np.asarray(orc.margins, np.float32), t_top1=np.asarray(orc.top1, np.float32),
                p_margin=np.asarray(orc.pred_margins, np.float32).reshape(FRAMES, 15),
                p_top1=np.asarray(orc.pred_top1, np.float32).reshape(FRAMES, 15))


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    out = {}
    for size, dtype, tag in (("0p6b", torch.bfloat16, "bf16"), ("0p6b", torch.float32, "f32"), ("1p7b", torch.bfloat16, "bf16")):
        r = run_case(size, dtype)
        for k, v in r.items():
            out[f"{size}_{tag}_{k}"] = v
        print(f"{size} {tag}: min talker margin {r['t_margin'].min():.4f}, min predictor margin {r['p_margin'].min():.4f}", flush=True)
    out["meta"] = np.asarray([FRAMES, PROMPT, TRAILING, SEED], np.int32)
    path = os.path.join(ROOT, "tests", "golden", "fulldepth_alt.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
