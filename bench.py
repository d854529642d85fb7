#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X: real-time factor (+ p50 TTFA) of
Qwen3-TTS-12Hz-0.6B voice-clone streaming (chunk_size=8), hipGraph decode, 1/2/4/8 GPUs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A *step* is one whole utterance per GPU through the hot path: synthetic 200-token ICL prompt
(170 reference frames) -> prefill -> exactly 200 generated frames (16 s of audio; EOS suppressed with
the API's own min_new_tokens knob) with product-default sampling, streamed in
8-frame chunks through the codec decoder with the reference's phase-1/phase-2 windowing.  Inputs
(weights, prompt embeddings) are resident in HBM before the timed region.  Utterances are sharded
over ranks (replicated weights, no data-path collective); RCCL is used for the barrier, the max-time
reduction and the final result gather only.  value = total audio seconds over all ranks / max wall.

The JSON line also carries:
  roofline      decode-frame hipGraph (one replay = one 80 ms frame): algorithmic bytes of SURVEY.md
                section 8(d) / measured replay time (HIP events on the launch stream) vs 8 TB/s
  cpu_baseline  the CPU oracle (oracle/, kind "port") timed on this box's host cores on a bounded
                sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

import numpy as np
import torch
import torch.distributed as dist

FRAMES = 200
PROMPT_LEN = 200
REF_FRAMES = 170
CHUNK = 8
FRAME_S = 1920 / 24000.0          # 12.5 frames per second


def algorithmic_bytes_per_frame(cfg, p: float) -> float:
    """every bf16 weight read once per frame + the p live KV rows."""
    t, pc = cfg.talker, cfg.predictor

    def stack(c):
        return c.num_hidden_layers * (c.hidden_size * c.q_dim + 2 * c.hidden_size * c.kv_dim + c.q_dim * c.hidden_size
                                      + 3 * c.hidden_size * c.intermediate_size)
    params = stack(t) + t.hidden_size * t.vocab_size + stack(pc) + (cfg.num_code_groups - 1) * pc.hidden_size * pc.vocab_size
    params += t.hidden_size * pc.hidden_size       # small_to_mtp projection (counted as in SURVEY.md)
    kv_row = t.num_hidden_layers * 2 * t.num_key_value_heads * t.head_dim * 2
    return 2.0 * params + kv_row * p


def build_model(device):
    from fq3hip.config import qwen3_tts_0p6b
    from fq3hip.weights import synth_weights, synth_prompt
    from fq3hip.model import FasterQwen3TTS
    cfg = qwen3_tts_0p6b()
    W = synth_weights(cfg, 0, torch.bfloat16, parts=("talker", "predictor", "codec"))
    model = FasterQwen3TTS.from_weights(cfg, W, device=device, dtype=torch.bfloat16, max_seq_len=2048,
                                        codec_max_frames=REF_FRAMES + FRAMES + 16, max_frames=FRAMES + 8)
    model._bench_weights = W
    return cfg, model


def concurrent_throughput(cfg, model, prompt, device, streams=4, utterances=2):
    """Extra (not the headline): S utterances in flight on ONE GPU, each with its own decode context, codec
    workspace, hipGraph and HIP stream, all borrowing the single weight replica.  Batch-1 decode leaves
    >90 % of the chip idle (latency-bound launches), so independent utterances overlap almost freely."""
    import threading
    from fq3hip.model import FasterQwen3TTS
    models = [model] + [FasterQwen3TTS.from_weights(cfg, model._bench_weights, device=device, dtype=torch.bfloat16,
                                                    max_seq_len=2048, codec_max_frames=REF_FRAMES + FRAMES + 16,
                                                    max_frames=FRAMES + 8, share=model) for _ in range(streams - 1)]
    bar = threading.Barrier(streams)
    res = [None] * streams

    errors = []

    def worker(i):
        try:
            torch.cuda.set_device(torch.device(device))
            st = torch.cuda.Stream(device=device)
            with torch.cuda.stream(st):
                one_utterance(models[i], prompt, 5000 + i, sync=st.synchronize)     # per-context warm-up + graph capture
                bar.wait(timeout=60)
                t0 = time.perf_counter()
                frames, ttfas = 0, []
                for u in range(utterances):
                    ttfa, wall, n, _ = one_utterance(models[i], prompt, 6000 + 10 * i + u, sync=st.synchronize)
                    frames += n; ttfas.append(ttfa)
                res[i] = (t0, time.perf_counter(), frames, ttfas)
        except BaseException as e:      # never leave the other workers parked on the barrier
            errors.append(repr(e))
            bar.abort()

    th = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(streams)]
    for t in th:
        t.start()
    deadline = time.time() + 150
    for t in th:
        t.join(timeout=max(1.0, deadline - time.time()))
    if errors or any(r is None for r in res):
        return {"streams": streams, "error": "; ".join(errors) or "worker timed out"}
    t0 = min(r[0] for r in res); t1 = max(r[1] for r in res)
    frames = sum(r[2] for r in res)
    ttfas = [x for r in res for x in r[3]]
    return {"streams": streams, "utterances": streams * utterances, "value": round(frames * FRAME_S / (t1 - t0), 3),
            "unit": "x real-time (aggregate audio s / wall s, one GPU)", "ttfa_ms_p50": round(1000 * float(np.median(ttfas)), 2)}


def batched_throughput(model, prompt, lanes=8, utterances=16):
    """Opt-in extra (--batch B, N=1 only): `utterances` synthetic utterances through `lanes` lock-step lanes
    (fq3_batch_*, continuous batching), vocoded as they finish.  Aggregate audio seconds / wall seconds."""
    from fq3hip.batching import BatchRequest
    tie, tam, tth, tpe, ref_codes = prompt
    m = model.model.model
    talker, config = m.talker, m.config.talker_config
    kw = model._gen_kwargs(FRAMES, FRAMES, 0.9, 50, 1.0, True, 1.05)
    dec = model._batch_decoder(lanes)
    reqs = [BatchRequest(i, talker, tie, tam, tth, tpe, config, dict(kw)) for i in range(utterances)]
    list(dec.run(reqs[:lanes]))                                  # warm-up: contexts, graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = 0
    for rid, codes, timing in dec.run(reqs):
        if codes is None:
            continue
        full = torch.cat([ref_codes.to(codes.device), codes], dim=0) if ref_codes is not None else codes
        audio_list, _sr = m.speech_tokenizer.decode({"audio_codes": full.unsqueeze(0)})
        _ = audio_list[0].cpu() if hasattr(audio_list[0], "cpu") else audio_list[0]
        frames += codes.shape[0]
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return {"lanes": lanes, "utterances": utterances, "value": round(frames * FRAME_S / wall, 3),
            "unit": "x real-time (aggregate audio s / wall s, one GPU, non-streaming vocoder per finished utterance)"}


def one_utterance(model, prompt, seed, sync=None):
    """Streaming voice-clone of one synthetic utterance.  Returns (ttfa_s, wall_s, n_frames, pcm)."""
    sync = sync or torch.cuda.synchronize
    tie, tam, tth, tpe, ref_codes = prompt
    m = model.model.model
    talker, config = m.talker, m.config.talker_config
    torch.manual_seed(seed)
    kw = model._gen_kwargs(FRAMES, FRAMES, 0.9, 50, 1.0, True, 1.05)
    sync()
    t0 = time.perf_counter()
    ttfa, chunks, frames = None, [], 0
    for audio, sr, timing in model._run_streaming(m, talker, config, tie, tam, tth, tpe, ref_codes, kw, CHUNK):
        if ttfa is None:
            ttfa = time.perf_counter() - t0       # `audio` is a host array: the first chunk is complete here
        chunks.append(audio)
        frames = timing["total_steps_so_far"]
    sync()
    wall = time.perf_counter() - t0
    return ttfa, wall, frames, np.concatenate(chunks) if chunks else np.zeros(0, np.float32)


def measure_frame_graph(model, prompt, n=64):
    """HIP events on the launch stream around n consecutive decode-frame graph replays."""
    from fq3hip.generate import _prefill_and_arm, run_frames
    tie, tam, tth, tpe, _ = prompt
    m = model.model.model
    eng, tn, pn, _ = _prefill_and_arm(m.talker, tie, tam, tth, tpe, m.config.talker_config, model.predictor_graph,
                                      model.talker_graph, FRAMES, FRAMES, 0.9, 50, 1.0, True, 1.05, use_graph=True)
    run_frames(eng, tn, pn, 0, 64)              # frames 0..63 (ring fill + warm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tn.exponential_(1); pn.exponential_(1)
    e0.record()
    eng.decode_frames(n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    p_mid = PROMPT_LEN + 64 + n / 2
    return ms, p_mid


def cpu_baseline(cfg, frames=6, budget_s=45.0):
    """CPU oracle ("port") on the host cores: same shapes, fp32 (bf16 matmuls are not accelerated on this
    host), same prompt; a BOUNDED sample: 200-token prefill + up to `frames` generated frames + their
    vocoding, cut short when `budget_s` is exceeded.  RTF with the same definition as the GPU line."""
    from fq3hip.weights import synth_weights, synth_prompt
    from oracle import qwen3tts_oracle as O
    cores = min(os.cpu_count() or 1, 16)        # batch-1 matvecs stop scaling (and regress) beyond ~16 threads
    torch.set_num_threads(cores)
    W = synth_weights(cfg, 0, torch.float32, parts=("talker", "predictor", "codec"))
    tie, tam, tth, tpe, ref = synth_prompt(cfg, PROMPT_LEN, 32, REF_FRAMES, dtype=torch.float32)
    orc = O.OracleTTS(cfg, W, max_seq_len=512)
    t0 = time.perf_counter()
    with torch.inference_mode():
        # frame-by-frame so the budget can stop the sample: first a 1-frame run (prefill + frame), then more
        sp = O.SamplingParams(max_new_tokens=1, min_new_tokens=1)
        codes = orc.generate(tie, tam, tth, tpe, sp)
        t1 = time.perf_counter() - t0
        per_frame_est = None
        if t1 < budget_s / 3:
            n_more = frames
            t2 = time.perf_counter()
            sp = O.SamplingParams(max_new_tokens=2, min_new_tokens=2)
            orc.generate(tie, tam, tth, tpe, sp)
            per_frame_est = max(time.perf_counter() - t2 - t1, 1e-3)      # one extra frame beyond the prefill+1 run
            n_more = int(max(1, min(frames, (budget_s - (time.perf_counter() - t0)) / (per_frame_est + 1e-9) / 2)))
            sp = O.SamplingParams(max_new_tokens=n_more, min_new_tokens=n_more)
            t3 = time.perf_counter()
            codes = orc.generate(tie, tam, tth, tpe, sp)
            t_codes = time.perf_counter() - t3
        else:
            t_codes = t1
        tv = time.perf_counter()
        wav = O.codec_decode(codes % cfg.codec.codebook_size, W, cfg.codec)
        t_voc = time.perf_counter() - tv
    n = codes.shape[0]
    wall = t_codes + t_voc
    return {"value": round(n * FRAME_S / wall, 5), "unit": "x real-time (audio s / wall s)", "cores": cores, "kind": "port",
            "sample": f"oracle/qwen3tts_oracle.py fp32 eager Torch, 0.6B shapes, {PROMPT_LEN}-token prefill + {n} frames "
                      f"+ vocoding of those frames ({t_codes:.1f}s + {t_voc:.1f}s); host has {os.cpu_count()} cores, {cores} threads used",
            "ms_per_frame": round(1000 * t_codes / n, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=0,
                    help="opt-in extra (N=1 only): lock-step lanes for the batched-decode throughput figure (0 = skip)")
    ap.add_argument("--concurrent", type=int, default=4,
                    help="extra figure (N=1 only, after the timed region): utterances in flight on one GPU (0 = skip)")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (there is no CPU path for the product)")
    # FQ3_BENCH_BACKEND=gloo + FQ3_BENCH_ONE_DEVICE=1: smoke-test the N>1 code path on a 1-GPU box
    backend = os.environ.get("FQ3_BENCH_BACKEND", "nccl")
    if os.environ.get("FQ3_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    coll_dev = device if backend == "nccl" else "cpu"

    from fq3hip.weights import synth_prompt
    cfg, model = build_model(device)
    prompt = [t.to(device) if t is not None else None
              for t in synth_prompt(cfg, PROMPT_LEN, 32, REF_FRAMES, dtype=torch.bfloat16)]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # (a high-priority decode stream was tried: no single-stream gain, and it quarters the throughput of the
    #  concurrent-utterance mode -- profiles/r01_concurrent_streams.txt -- so everything stays on default-priority streams)
    for i in range(args.warmup):
        one_utterance(model, prompt, 1000 + i)
    frame_ms, p_mid = measure_frame_graph(model, prompt)

    barrier()
    t0 = time.perf_counter()
    ttfas, rtfs, frames_total, pcm = [], [], 0, None
    for i in range(args.steps):
        ttfa, wall, n, pcm = one_utterance(model, prompt, 2000 + rank * 100 + i)
        ttfas.append(ttfa); rtfs.append(n * FRAME_S / wall); frames_total += n
    barrier()
    elapsed = time.perf_counter() - t0

    conc = None
    if args.concurrent > 1 and world == 1:
        try:
            conc = concurrent_throughput(cfg, model, prompt, device, streams=args.concurrent)
        except Exception as e:      # an extra; never lose the headline line to it
            conc = {"error": repr(e)}

    batched = None
    if args.batch > 1 and world == 1:
        try:
            batched = batched_throughput(model, prompt, lanes=min(args.batch, 8), utterances=2 * min(args.batch, 8))
        except Exception as e:      # an extra; never lose the headline line to it
            batched = {"error": repr(e)}

    # max over ranks of the wall time, sum of frames, gather of TTFAs and (result gather) PCM lengths
    if world > 1:
        from fq3hip.sharding import gather_arrays
        tt = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
        ft = torch.tensor([frames_total], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(ft)
        frames_total = int(ft)
        all_ttfa = gather_arrays(np.asarray(ttfas, dtype=np.float32), coll_dev)
        ttfas = [float(x) for a in all_ttfa for x in a]
        gathered = gather_arrays(pcm.astype(np.float32), coll_dev)     # the result gather of the north star
        n_gathered = sum(len(a) for a in gathered)
    else:
        n_gathered = len(pcm)

    if rank == 0:
        audio_s = frames_total * FRAME_S
        value = audio_s / elapsed
        bytes_frame = algorithmic_bytes_per_frame(cfg, p_mid)
        achieved = bytes_frame / (frame_ms * 1e-3) / 1e9
        out = {
            "metric": "real-time factor (audio s / wall s), Qwen3-TTS-12Hz-0.6B voice-clone streaming chunk_size=8; p50 TTFA in ttfa_ms_p50",
            "value": round(value, 3), "unit": "x real-time", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / max(args.steps, 1), 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (seeded random weights at 0.6B shapes, synthetic 200-token ICL prompt)",
            "config": {"workload": "configs[1]: Qwen3-TTS-12Hz-0.6B-Base voice-clone streaming chunk_size=8, hipGraph decode",
                       "prompt_tokens": PROMPT_LEN, "ref_frames": REF_FRAMES, "frames_per_utterance": FRAMES,
                       "timed_region": "prefill + decode + streaming vocoder; prompt embeddings are the (HBM-resident) input, "
                                       "text tokenisation / prompt assembly is host glue outside the hot path",
                       "utterances_per_gpu": args.steps, "sampling": "T=0.9 top_k=50 top_p=1.0 rep=1.05 (predictor T=0.9 top_k=50)",
                       "parallelism": f"utterance-sharded x{world} (replicas, result gather only)"},
            "ttfa_ms_p50": round(1000 * float(np.median(ttfas)), 2), "ttfa_ms_mean": round(1000 * float(np.mean(ttfas)), 2),
            "rtf_single_stream_mean": round(float(np.mean(rtfs)), 3),
            "decode_ms_per_frame": round(frame_ms, 4), "gathered_samples": int(n_gathered),
            "reference_published": {"rtx4090_rtf": 4.78, "rtx4090_ttfa_ms": 156, "h100_rtf": 3.884, "h100_ttfa_ms": 228,
                                    "source": "reference README.md:227,229 (CUDA graphs, other hardware)"},
            "roofline": {"bound": "hbm", "kernel": "decode-frame hipGraph (554 launches: predictor M=2 prefill pass + 14 token passes, talker 28 layers, heads, samplers)",
                         "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                         "algorithmic_bytes_per_launch": int(bytes_frame), "kv_len": p_mid,
                         # measured once per round with a separate `rocprofv3 --pmc FETCH_SIZE` pass (x2 gfx950 correction):
                         # the 15 predictor weight passes re-read their 157 MB through the fabric every pass
                         "traffic": 3.37e9, "traffic_source": "profiles/r01_pmc_fetch_size.txt",
                         "launch_ms": round(frame_ms, 4)},
        }
        if conc is not None:
            out["concurrent_utterances_one_gpu"] = conc
        if batched is not None:
            out["batched_decode_one_gpu"] = batched
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:          # the baseline is a reported extra; never lose the GPU line to it
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
