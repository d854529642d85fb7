#!/usr/bin/env python3
"""Batched decode timing (development aid): B lanes of the 0.6B / 1.7B-shape model in lock-step, one hipGraph per frame.
usage: batch_bench.py [size=0.6b|1.7b] [B list, e.g. 8,16] [frames=48] [graph=1|0] [skinny=-|0|1|2 -> fq3_batch_set_option("skinny", v)]
                      [groups list, e.g. 1,2,4 -> fq3_batch_set_option("groups", g); the codes of every lane are compared across the list]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))
import torch
from fq3hip.config import qwen3_tts_0p6b, qwen3_tts_1p7b
from fq3hip.weights import synth_weights, synth_prompt
from fq3hip.engine import Fq3Engine, Fq3Batch


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "0.6b"
    Bs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8").split(",")]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 48
    graph = (sys.argv[4] if len(sys.argv) > 4 else "1") != "0"
    skinny = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] != "-" else None
    groups = [int(x) for x in sys.argv[6].split(",")] if len(sys.argv) > 6 else [None]
    # FQ3_BENCH_SWEEP="norm_fused=0;norm_fused=1;norm_fused=1,norm_skinny_above=8": option sets measured one after another in this process
    # (instead of the groups list): every lane count x every set
    sweep = [v for v in os.environ.get("FQ3_BENCH_SWEEP", "").split(";") if v]
    if sweep:
        groups = sweep
    cfg = qwen3_tts_0p6b() if size == "0.6b" else qwen3_tts_1p7b()
    dt = torch.bfloat16
    W = synth_weights(cfg, 0, dt, parts=("talker", "predictor"))
    first = Fq3Engine(cfg, W, "cuda", dt, max_seq_len=1024, max_frames=256)
    lanes = [first] + [Fq3Engine(cfg, W, "cuda", dt, max_seq_len=1024, max_frames=256, share=first) for _ in range(max(Bs) - 1)]
    V, Vp = cfg.talker.vocab_size, cfg.predictor.vocab_size
    kw = dict(temperature=0.9, top_k=50, top_p=1.0, do_sample=True)
    nf = 64
    for B in Bs:
      ref_codes = None
      for G in groups:
        torch.manual_seed(1000 + B)
        keep = []
        for i, eng in enumerate(lanes[:B]):
            tie, tam, tth, tpe, _ = synth_prompt(cfg, 200, 32, 0, dtype=dt, seed=1234 + 10 * i)
            logits, hidden = eng.prefill(tie[0].cuda().contiguous())
            tn = torch.empty(nf, V, dtype=dt, device="cuda").exponential_(1)
            pn = torch.empty(nf, 15, Vp, dtype=dt, device="cuda").exponential_(1)
            fn = torch.empty(V, dtype=dt, device="cuda").exponential_(1)
            tok = eng.sample(logits, sup_lo=V - 1024, sup_hi=V, keep_id=cfg.codec_eos_token_id, suppress_eos=True, noise=fn, **kw)
            eng.decode_begin(first_token=int(tok), prefill_len=200, gen_step=0, past_hidden=hidden,
                             trailing_text=tth[0].cuda().contiguous(), tts_pad_embed=tpe.view(-1).cuda().contiguous(),
                             repetition_penalty=1.05, min_new_tokens=250, max_new_tokens=250,
                             talker_noise=tn, pred_noise=pn, noise_frames=nf, **kw)
            keep.append((tn, pn))
        batch = Fq3Batch(lanes[:B])
        if skinny is not None:
            batch.set_option("skinny", skinny)
        if isinstance(G, str):
            for kv in G.split(","):
                k, v = kv.split("=")
                batch.set_option(k, int(v))
        elif G is not None:
            batch.set_option("groups", G)
        if os.environ.get("FQ3_BENCH_NORM_SKINNY_ABOVE") is not None:
            batch.set_option("norm_skinny_above", int(os.environ["FQ3_BENCH_NORM_SKINNY_ABOVE"]))
        if os.environ.get("FQ3_BENCH_NORM_SKINNY") is not None:       # A/B: 0 = the panel kernels above 64 lanes as well
            batch.set_option("norm_skinny", int(os.environ["FQ3_BENCH_NORM_SKINNY"]))
        if os.environ.get("FQ3_BENCH_NORM_FUSED") is not None:        # A/B: 0 = the separate normalisation launch (round-4 form)
            batch.set_option("norm_fused", int(os.environ["FQ3_BENCH_NORM_FUSED"]))
        for kv in os.environ.get("FQ3_BENCH_OPTS", "").split(","):    # further A/B switches: key=value[,key=value]
            if "=" in kv:
                k, v = kv.split("=")
                batch.set_option(k, int(v))
        if graph:
            batch.graph_capture()
        batch.frames(8); torch.cuda.synchronize()
        t0 = time.perf_counter(); batch.frames(frames); torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / frames
        n = [e.decode_poll()[0] for e in lanes[:B]]
        same = ""
        if len(groups) > 1 and not sweep:
            codes = [e.decode_codes(0, min(n)).cpu() for e in lanes[:B]]
            if ref_codes is None:
                ref_codes = codes
            else:
                same = f", codes of all {B} lanes == groups={groups[0]}: {all(torch.equal(a, b) for a, b in zip(codes, ref_codes))}"
        print(f"{size} B={B} graph={int(graph)}{'' if skinny is None else f' skinny={skinny}'}{'' if G is None else (f' [{G}]' if isinstance(G, str) else f' groups={G}')}: {ms:.3f} ms per lock-step frame -> {B * 80.0 / ms:.1f}x real-time aggregate "
              f"({80.0 / ms:.1f}x per lane), frames per lane {sorted(set(n))}{same}", flush=True)
        batch.close()


if __name__ == "__main__":
    main()
