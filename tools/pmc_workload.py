#!/usr/bin/env python3
"""Small fixed workloads for `rocprofv3 --pmc` passes (development aid; bench.py is the contract).
usage: pmc_workload.py codec | prefill | frames | batch"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def main():
    what = sys.argv[1]
    dev = "cuda:0"
    cfg, model = bench.build_model(dev)
    req = bench.build_request(cfg, dev)
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
        dec = model._batch_decoder(8)
        keep = [_prefill_and_arm(m.talker, tie, tam, tth, tpe, m.config.talker_config, ln.predictor_graph, ln.talker_graph,
                                 200, 200, 0.9, 50, 1.0, True, 1.05, use_graph=False) for ln in dec.lanes]
        for _e, tn, pn, _ in keep:
            tn.exponential_(1); pn.exponential_(1)
        dec.batch.graph_reset()
        dec.batch.frames(8)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
