#!/usr/bin/env python3
"""Teacher-forced decision counts of the lock-step batch under option sets that are NOT the default at a lane count (development aid:
what enabling a kernel form at more lane counts would do to the per-lane counts).  Reuses the harness of tests/test_gpu_batch_fulldepth.py.
usage: parity_probe_batch.py [0p6b|1p7b] [lanes list] [option sets, e.g. "attn_lane=2;attn_lane=2,pred_pair=1,norm_skinny_above=8,skinny=2"]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_batch_fulldepth as T
from fq3hip.config import qwen3_tts_0p6b, qwen3_tts_1p7b
from fq3hip.weights import synth_weights
from fq3hip.engine import Fq3Engine


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "0p6b"
    lanes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16,32").split(",")]
    sets = [s for s in (sys.argv[3] if len(sys.argv) > 3 else "attn_lane=2").split(";") if s]
    golden = os.path.join(ROOT, "tests", "golden")
    cfg = qwen3_tts_0p6b() if size == "0p6b" else qwen3_tts_1p7b()
    dtype = torch.bfloat16
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    cases = T._cases(golden, cfg, size, "bf16", dtype)
    seq = max(c[1].shape[1] + c[0]["codes"].shape[0] for c in cases) + 8
    frames = max(c[0]["codes"].shape[0] for c in cases) + 8
    first = Fq3Engine(cfg, W, device="cuda", dtype=dtype, max_seq_len=seq, max_frames=frames)
    engines = [first] + [Fq3Engine(cfg, first.weights, device="cuda", dtype=dtype, max_seq_len=seq, max_frames=frames, share=first) for _ in range(max(lanes) - 1)]
    for e in engines:
        e.set_predictor_sampling(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    print("oracle-derived floor:", T.oracle_floor(golden, size))
    for B in lanes:
        for s in ["default"] + sets:
            opts = () if s == "default" else tuple((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(","))
            sc = T._run_batch(engines, cfg, cases, B, mfma=1, options=opts)
            print(f"{size} B={B} [{s}]: per lane {[x['matched_decisions'] for x in sc[:2]]}, unexplained {sum(x['unexplained'] for x in sc)}, "
                  f"worst mismatch {max(x['worst_mismatch_ulp'] for x in sc)} ulps", flush=True)


if __name__ == "__main__":
    main()
