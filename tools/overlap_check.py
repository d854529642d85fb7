#!/usr/bin/env python3
"""Where does the end-to-end utterance time go?  (development aid)  Same utterance as bench.py, timed
  A  streaming generator only (codes, no vocoder)
  B  product path: vocoder on its own stream
  C  B with the decode loop on a high-priority stream
  D  B with FQ3_VOC_PRIORITY=<lowest> (set in the environment before launch)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))
import numpy as np, torch
import bench

def codes_only(model, prompt, seed):
    from fq3hip.streaming import fast_generate_streaming
    tie, tam, tth, tpe, ref = prompt
    m = model.model.model
    torch.manual_seed(seed)
    kw = model._gen_kwargs(bench.FRAMES, bench.FRAMES, 0.9, 50, 1.0, True, 1.05)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for chunk, timing in fast_generate_streaming(talker=m.talker, talker_input_embeds=tie, attention_mask=tam, trailing_text_hiddens=tth,
                                                 tts_pad_embed=tpe, config=m.config.talker_config, predictor_graph=model.predictor_graph,
                                                 talker_graph=model.talker_graph, chunk_size=bench.CHUNK, **kw):
        n = timing["total_steps_so_far"]
    torch.cuda.synchronize()
    return time.perf_counter() - t0, n

def main():
    print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a",
          " FQ3_VOC_PRIORITY =", os.environ.get("FQ3_VOC_PRIORITY"))
    from fq3hip.weights import synth_prompt
    cfg, model = bench.build_model("cuda")
    prompt = [t.to("cuda") if t is not None else None
              for t in synth_prompt(cfg, bench.PROMPT_LEN, 32, bench.REF_FRAMES, dtype=torch.bfloat16)]
    for i in range(2): bench.one_utterance(model, prompt, 100 + i)
    a = [codes_only(model, prompt, 200 + i)[0] for i in range(4)]
    b = [bench.one_utterance(model, prompt, 300 + i) for i in range(4)]
    hp = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(hp):
        c = [bench.one_utterance(model, prompt, 400 + i, sync=lambda: torch.cuda.synchronize()) for i in range(4)]
    f = lambda xs: f"{1e3 * float(np.median(xs)):.1f} ms"
    print("A codes only           :", f(a), " (200 frames)")
    print("B product (side stream):", f([x[1] for x in b]), " ttfa", f([x[0] for x in b]))
    print("C decode on hi-priority:", f([x[1] for x in c]), " ttfa", f([x[0] for x in c]))

if __name__ == "__main__":
    main()
