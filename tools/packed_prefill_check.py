#!/usr/bin/env python3
"""Decision measurement for BatchDecoder(packed_prefill=...): first-wave TTFA and aggregate RTF of the batched streaming entry
point with one prefill per request vs ONE packed prefill per admitted group (fq3_prefill_batch), workspaces reserved up front.
usage: packed_prefill_check.py [lanes=8,16]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def main():
    lanes_list = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,16").split(",")]
    dev = "cuda:0"
    cfg, model = bench.build_model(dev)
    req = bench.build_request(cfg, dev)
    bench.one_utterance(model, req, 1, frames=16)
    for lanes in lanes_list:
        for packed in (False, True, False, True):
            dec = model._batch_decoder(lanes)
            dec.packed_prefill = packed
            bench.batched_streaming_run(model, req, lanes, lanes)                   # warm-up
            r = bench.batched_streaming_run(model, req, lanes, 2 * lanes)
            print(json.dumps({"lanes": lanes, "packed_prefill": packed, "aggregate_rtf": r["value"],
                              "ttfa_first_wave_p50_ms": r["ttfa_ms_first_wave_p50"], "ttfa_first_wave_max_ms": r["ttfa_ms_first_wave_max"]}), flush=True)
        model._batch_decoder(lanes).packed_prefill = False


if __name__ == "__main__":
    main()
