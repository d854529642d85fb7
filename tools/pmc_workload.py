#!/usr/bin/env python3
"""Small fixed workloads for `rocprofv3 --pmc` passes (development aid; bench.py is the contract).
usage: pmc_workload.py codec | prefill | prefill4k | frames | batch [lanes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def main():
    what = sys.argv[1]
    dev = "cuda:0"
    if what == "prefill4k":
        # BASELINE configs[4] shape: a 4096-row prompt through the matrix-core prefill at the 1.7B dims (flash attention + 256-wide tiles)
        from fq3hip.weights import synth_prompt
        cfg, model = bench.build_model(dev, "1p7b", max_seq_len=6144)
        x = (synth_prompt(cfg, 4096, 4, 0, dtype=torch.bfloat16)[0][0] * 30).to(torch.bfloat16).to(dev).contiguous()
        eng = model.talker_graph.engine
        for _ in range(2):
            eng.prefill(x)
        torch.cuda.synchronize()
        return
    cfg, model = bench.build_model(dev)
    req = bench.build_request(cfg, dev)
    if what not in ("frames", "batch"):                     # the counter passes over decode frames run direct launches only
        bench.one_utterance(model, req, 1, frames=16)
    prompt = bench.prepared_prompt(model, req)
    torch.cuda.synchronize()
    if what == "codec":
        tok = model.model.model.speech_tokenizer
        g = torch.Generator().manual_seed(4)
        codes = torch.randint(0, cfg.codec.codebook_size, (370, 16), generator=g).to(dev)
        for _ in range(2):
            tok.decode_tensor(codes)
    elif what == "prefill":
        eng = model.talker_graph.engine
        for _ in range(3):
            eng.prefill(prompt[0][0].contiguous())
    elif what == "frames":
        # direct launches (rocprofv3 --pmc crashes on hipGraph replays on this stack): 8 frames at KV ~ 230
        from fq3hip.generate import _prefill_and_arm, run_frames
        tie, tam, tth, tpe, _ = prompt
        m = model.model.model
        eng, tn, pn, _ = _prefill_and_arm(m.talker, tie, tam, tth, tpe, m.config.talker_config, model.predictor_graph,
                                          model.talker_graph, 200, 200, 0.9, 50, 1.0, True, 1.05, use_graph=False)
        run_frames(eng, tn, pn, 0, 8)
    elif what == "batch":
        from fq3hip.generate import _prefill_and_arm
        tie, tam, tth, tpe, _ = prompt
        m = model.model.model
        dec = model._batch_decoder(int(sys.argv[2]) if len(sys.argv) > 2 else 16)
        keep = [_prefill_and_arm(m.talker, tie, tam, tth, tpe, m.config.talker_config, ln.predictor_graph, ln.talker_graph,
                                 200, 200, 0.9, 50, 1.0, True, 1.05, use_graph=False) for ln in dec.lanes]
        for _e, tn, pn, _ in keep:
            tn.exponential_(1); pn.exponential_(1)
        dec.batch.graph_reset()
        dec.batch.frames(8)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
