#!/usr/bin/env python3
"""End-to-end batched throughput A/B (development aid): bench.batched_run -- the product's _run_batch_full -- at a lane count, for
several `batch_vocode_every` settings (0 = vocode when an utterance ends; N = exact slices every N frames while it decodes).
usage: batch_e2e_bench.py [0p6b|1p7b] [lanes=64] [every list, e.g. 0,64,100] [codec=bf16x2] [groups list, e.g. 1,2: model.batch_groups | -] [timed runs=2] [decode stream priority: 0 = the current stream | -1 = a high-priority stream]
                          [lookahead list, e.g. 1,0: model.batch_lookahead]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "0p6b"
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    everys = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,64,100").split(",")]
    codec = sys.argv[4] if len(sys.argv) > 4 else bench.HEADLINE_CODEC
    groups = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 and sys.argv[5] != "-" else [None]
    runs = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    prio = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    las = [int(x) for x in sys.argv[8].split(",")] if len(sys.argv) > 8 else [None]
    dev = "cuda:0"
    if prio:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=prio))
    cfg, model = bench.build_model(dev, size, max_seq_len=2048, codec_precision=codec)
    model.batch_kv_blocks = max(model.batch_kv_blocks, 2 * lanes * ((bench.PROMPT_LEN + bench.FRAMES + 64 + 63) // 64))   # lanes + as many spare contexts
    req = bench.build_request(cfg, dev)
    prompt = bench.prepared_prompt(model, req)
    bench.batched_run(model, prompt, lanes, lanes)
    for G, la in [(g, l) for g in groups for l in las]:
      if G is not None:
        model.batch_groups = G
      if la is not None:
        model.batch_lookahead = la
      # FQ3_E2E_FIRST_WAVE="0,32": model.batch_first_wave (requests armed before the first frame is queued; 0 = one per lane)
      fws = [int(x) for x in os.environ.get("FQ3_E2E_FIRST_WAVE", "0").split(",")]
      # FQ3_E2E_VOC_SHARE="0,0.5,0.75": model.vocoder_cu_share (the vocoder stream confined to a share of the CUs; 0 = the whole chip)
      shares = [float(x) for x in os.environ.get("FQ3_E2E_VOC_SHARE", "0").split(",")]
      for ev, fw, sh in [(e, f, s_) for s_ in shares for e in everys for f in fws]:
        if len(shares) > 1 or sh > 0:
            model.vocoder_cu_share = sh if sh > 0 else None
            model._voc_stream = None; model._side_voc_stream = None          # the next run builds its vocoder stream anew
            print(f"  vocoder_cu_share = {sh}", flush=True)
        model.batch_vocode_every = ev
        model.batch_first_wave = fw if fw > 0 else None
        bench.batched_run(model, prompt, lanes, lanes)
        res = [bench.batched_run(model, prompt, 2 * lanes, lanes, seed0=2000 + i) for i in range(runs)]
        host = getattr(model._batch_cache[1], "host_s", None) if getattr(model, "_batch_cache", None) else None
        if host:
            print("  scheduler thread, last run (s): " + ", ".join(f"{k} {v:.3f}" for k, v in host.items()), flush=True)
        print(f"{size} lanes={lanes} codec={codec} batch_vocode_every={ev} first_wave={fw}{'' if G is None else f' groups={G}'}{'' if la is None else f' lookahead={la}'}: {[round(a / w, 1) for a, w, _l in res]} x real-time end to end "
              f"({2 * lanes} utterances, wall {[round(w, 3) for _a, w, _l in res]} s)", flush=True)


if __name__ == "__main__":
    main()
