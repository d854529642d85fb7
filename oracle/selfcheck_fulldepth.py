#!/usr/bin/env python3
"""How reproducible are the ORACLE's own bf16 ids under a change of summation order?

    python oracle/selfcheck_fulldepth.py [0p6b|1p7b]      # -> tests/golden/fulldepth_selfcheck.json (minutes on CPU)
    python oracle/selfcheck_fulldepth.py 1p7b_4096        # the configs[4] goldens (tests/golden/longprompt_full.npz; ~15 minutes)
    python oracle/selfcheck_fulldepth.py 0p6b_alt|1p7b_alt   # the second golden utterance (tests/golden/fulldepth_alt.npz: 137 rows, 16 frames)

The teacher-forced GPU parity tests (tests/test_gpu_fulldepth.py) accept a bf16 mismatch only where the oracle's own top-2 margin
is a few bf16 ulps, arguing that such a decision is made by the summation order inside dot products and not by the algorithm.
This script measures that claim on the oracle itself: the same model, prompt and golden history (tests/golden/fulldepth.npz), the
same algorithm and rounding points, but every F.linear evaluated with fp32 operands and one rounding of the result to bf16
instead of the CPU's native bf16 kernel (oneDNN blocking) -- i.e. only the order and width of the accumulation inside each dot
product changes, exactly what separates the HIP kernels from the oracle.  Every decision is scored by teacher forcing (the variant
records its own arg-max and continues with the golden id), like the GPU run.  The result -- how many of the 384 decisions flip, and
at which oracle margin the worst flip sits -- is the floor below which "bit-identical bf16 ids" cannot be asked of ANY second
implementation.  Test infrastructure only.
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

from fq3hip.config import qwen3_tts_0p6b, qwen3_tts_1p7b          # noqa: E402
from fq3hip.weights import synth_weights, synth_prompt            # noqa: E402
from oracle import qwen3tts_oracle as O                           # noqa: E402
from oracle import teacher_forced as TF                           # noqa: E402


class ForcedOracle(O.OracleTTS):
    """The oracle's decode loop with the golden ids forced after every decision (oracle/qwen3tts_oracle.py generate /
    predictor_loop, same calls in the same order); `decisions` collects this run's own arg-max ids."""

    def forced(self, tie, tam, tth, tpe, golden: torch.Tensor):
        cfg = self.cfg
        eos = cfg.codec_eos_token_id
        sm = O.build_suppress_mask(cfg.talker.vocab_size, eos)
        frames = golden.shape[0]
        dec = np.zeros((frames, 16), np.int64)

        def pick(logits):
            l = logits.detach().float().view(-1).clone()
            l[sm] = float("-inf")
            l[eos] = float("-inf")                    # min_new_tokens = frames: EOS suppressed at every step, as in the golden run
            return int(l.argmax())

        logits, past_hidden, gen_step, prefill_len = self.prefill(tie, tam)
        tth_d, tpe_d = tth.to(self.dtype), tpe.to(self.dtype)
        pre = "talker.code_predictor"
        for step in range(frames):
            dec[step, 0] = pick(logits)
            token = golden[step, 0].view(1)
            last = F.embedding(token.view(1, 1), self.W["talker.model.codec_embedding.weight"])
            pred_in = torch.cat((past_hidden, last), dim=1)
            # predictor, forced
            self.pcache.length = 0
            h = self._proj(pred_in[0].to(self.dtype))
            h = O.stack_forward(self.W, f"{pre}.model", cfg.predictor, h, 0, self.pcache, torch.arange(2).float())
            lg = F.linear(h[-1:], self.W[f"{pre}.lm_head.0.weight"])
            dec[step, 1] = int(lg.float().argmax())
            for cb in range(1, cfg.num_code_groups - 1):
                tok = golden[step, cb].view(1)
                emb = self._proj(F.embedding(tok, self.W[f"{pre}.model.codec_embedding.{cb - 1}.weight"]))
                h = O.stack_forward(self.W, f"{pre}.model", cfg.predictor, emb, 1 + cb, self.pcache, torch.tensor([1.0 + cb]))
                lg = F.linear(h[-1:], self.W[f"{pre}.lm_head.{cb}.weight"])
                dec[step, 1 + cb] = int(lg.float().argmax())
            c15 = golden[step, 1:]
            text_add = tth_d[:, gen_step].unsqueeze(1) if gen_step < tth_d.shape[1] else tpe_d
            x = self.embed_frame(token, c15, text_add)
            hidden = self.talker_step(x, prefill_len + step)
            logits = self.codec_head(hidden).view(1, 1, -1)
            past_hidden = hidden.clone()
            gen_step += 1
        return dec


class Fp32Linear:
    """Context: F.linear on bf16 operands is evaluated with wider operands and rounded once to bf16.
    mode "fp32": fp32 operands, the BLAS's own blocking; "fp64": float64 operands (the exactly rounded dot product, for all practical
    purposes); "ksplit8": fp32 operands, K cut into 8 slices whose partial products are summed in slice order -- the shape of the
    HIP prefill GEMMs' accumulation (8 waves split K)."""

    def __init__(self, mode: str = "fp32"):
        self.mode = mode

    def __enter__(self):
        self.orig = F.linear
        cache = {}
        wide = torch.float64 if self.mode == "fp64" else torch.float32

        def lin(x, w, b=None):
            if x.dtype != torch.bfloat16:
                return self.orig(x, w, b)
            key = w.data_ptr()
            if key not in cache:
                cache[key] = w.to(wide)
            xw, ww = x.to(wide), cache[key]
            if self.mode == "ksplit8" and ww.shape[1] % 8 == 0:
                ks = ww.shape[1] // 8
                y = self.orig(xw[..., :ks], ww[:, :ks])
                for i in range(1, 8):
                    y = y + self.orig(xw[..., i * ks:(i + 1) * ks], ww[:, i * ks:(i + 1) * ks])
            else:
                y = self.orig(xw, ww)
            if b is not None:
                y = y + b.to(wide)
            return y.to(torch.bfloat16)
        F.linear = lin
        torch.nn.functional.linear = lin
        return self

    def __exit__(self, *a):
        F.linear = self.orig
        torch.nn.functional.linear = self.orig


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "0p6b"
    size = key.split("_")[0]
    torch.set_num_threads(os.cpu_count() or 1)
    alt = key.endswith("_alt")
    g = np.load(os.path.join(ROOT, "tests", "golden", "longprompt_full.npz" if key.endswith("_4096") else ("fulldepth_alt.npz" if alt else "fulldepth.npz")))
    frames, plen, tlen = (int(x) for x in g["meta"][:3])
    case = TF.load_case(g, f"{size}_bf16")
    golden = torch.from_numpy(case["codes"].astype(np.int64))
    cfg = qwen3_tts_0p6b() if size == "0p6b" else qwen3_tts_1p7b()
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("talker", "predictor"))
    if alt:
        from oracle.make_golden_fulldepth_alt import alt_prompt
        tie, tam, tth, tpe, _ = alt_prompt(cfg, torch.bfloat16)
    else:
        tie, tam, tth, tpe, _ = synth_prompt(cfg, plen, tlen, 0, dtype=torch.bfloat16)
    out = {}
    for name in ("native_bf16_again", "native_bf16_one_thread", "fp32_operands", "fp32_operands_ksplit8", "fp64_operands"):
        orc = ForcedOracle(cfg, W, max_seq_len=plen + frames + 8)
        t0 = time.time()
        torch.set_num_threads(1 if name == "native_bf16_one_thread" else (os.cpu_count() or 1))
        with torch.inference_mode():
            if name.startswith("fp"):
                with Fp32Linear({"fp32_operands": "fp32", "fp32_operands_ksplit8": "ksplit8", "fp64_operands": "fp64"}[name]):
                    dec = orc.forced(tie, tam, tth, tpe, golden)
            else:
                dec = orc.forced(tie, tam, tth, tpe, golden)
        s = TF.score(dec, case, 3.0)
        s["seconds"] = round(time.time() - t0, 1)
        out[name] = s
        print(key, name, s, flush=True)
    path = os.path.join(ROOT, "tests", "golden", "fulldepth_selfcheck.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[key] = out
    json.dump(cur, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
