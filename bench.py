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
reduction and ONE final result gather.  value = total audio seconds over all ranks / max wall.

The JSON line also carries (all measured in this run unless tagged otherwise):
  roofline            decode-frame hipGraph (one replay = one 80 ms frame): algorithmic bytes of SURVEY.md
                      section 8(d) / measured replay time (HIP events on the launch stream) vs 8 TB/s
  roofline_mfma       the matrix-core kernels of the path (codec conv GEMMs, prefill GEMMs): FLOPs / event time vs the
                      2.5 PFLOP/s dense bf16 peak
  parity_bf16_frames  teacher-forced id agreement of this very model (bf16, full depth) with the CPU oracle's golden ids
  parity_pcm          PCM RMS of this model's codec (bf16, and the fp32 high-precision mode with its cost in ms) against the
                      fp32-arithmetic oracle's golden waveform on the same bf16 checkpoint weights
  batched_decode_one_gpu   32 lock-step lanes over one weight stream (fq3_batch_*; frame times at 8 / 16 lanes beside it), with its own HBM roofline
  config3_sharded_batched  BASELINE configs[3]: 1.7B-CustomVoice shapes, 64 utterances through generate_custom_voice_batch,
                      sharded over the ranks, up to 32 lanes per GPU (all ranks)
  model_1p7b          BASELINE configs[2]: the 1.7B shapes, single stream (N = 1 only); config4_voice_design_4k inside it:
                      BASELINE configs[4], a 4096-row VoiceDesign prompt streamed end to end
  cpu_baseline        the CPU oracle (oracle/, kind "port") timed on this box's host cores on a bounded
                      sample of the same workload (rank 0, N=1 only)

`--stub` (tests/test_bench_multirank.py): no GPU, a scripted utterance runner -- exercises exactly the N > 1 branch
(process group, barrier, max/sum reductions, the result gather) under gloo on CPU.
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
HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0    # dense (MI355X_MICROARCH.md; the 2:1-sparsity headline figure is never used)


def algorithmic_bytes_per_frame(cfg, p: float, lanes: int = 1) -> float:
    """every bf16 weight read once per frame + the p live KV rows (per lane)."""
    t, pc = cfg.talker, cfg.predictor

    def stack(c):
        return c.num_hidden_layers * (c.hidden_size * c.q_dim + 2 * c.hidden_size * c.kv_dim + c.q_dim * c.hidden_size
                                      + 3 * c.hidden_size * c.intermediate_size)
    params = stack(t) + t.hidden_size * t.vocab_size + stack(pc) + (cfg.num_code_groups - 1) * pc.hidden_size * pc.vocab_size
    params += t.hidden_size * pc.hidden_size       # small_to_mtp projection (counted as in SURVEY.md)
    kv_row = t.num_hidden_layers * 2 * t.num_key_value_heads * t.head_dim * 2
    return 2.0 * params + kv_row * p * lanes


def codec_gemm_flops(c, T: int) -> float:
    """FLOPs of the conv / linear contractions of one codec decode of T frames (2 * M * N * K per GEMM)."""
    f = 0.0
    f += 2 * T * c.rvq_dim * c.codebook_dim * 2 + 2 * T * 3 * c.codebook_dim * c.latent_dim
    QD = c.num_attention_heads * c.head_dim
    f += 2 * T * c.latent_dim * c.hidden_size * 2
    f += c.num_hidden_layers * 2 * T * (c.hidden_size * 3 * QD + QD * c.hidden_size + 3 * c.hidden_size * c.intermediate_size)
    rows = T
    for r in c.upsampling_ratios:
        f += 2 * rows * c.latent_dim * r * c.latent_dim
        rows *= r
        f += 2 * rows * c.latent_dim * 4 * c.latent_dim * 2
    ch = c.decoder_dim
    f += 2 * rows * 7 * c.latent_dim * ch
    for r in c.upsample_rates:
        f += 2 * (rows - 1) * 2 * ch * r * (ch // 2)
        rows = (rows - 1) * r
        ch //= 2
        f += 3 * (2 * rows * 7 * ch * ch + 2 * rows * ch * ch)
    return f


def prefill_gemm_flops(cfg, L: int) -> float:
    t = cfg.talker
    per_layer = t.hidden_size * (t.q_dim + 2 * t.kv_dim) + t.q_dim * t.hidden_size + 3 * t.hidden_size * t.intermediate_size
    return 2.0 * L * per_layer * t.num_hidden_layers


# ------------------------------------------------------------------------------------------------------------------
# GPU runner
# ------------------------------------------------------------------------------------------------------------------
SPEAKER = "synthetic"


HEADLINE_CODEC = "bf16x2"         # vocoder arithmetic of EVERY figure of this line (headline, batched, configs[2..4]): bf16 weights, activations as
                                  # bf16 high part + residual, two bf16 MFMAs per product -- the mode that meets the north star's 1e-3 PCM bound
                                  # (parity_pcm: ~1e-5) at ~1.9x the bf16 decode time; the bf16-codec figures are reported beside the headline


BATCH_LOOKAHEAD = None


def build_model(device, size="0p6b", frames=FRAMES, max_seq_len=2048, model_type="base", codec_precision=None, share=None):
    """`share`: a model built before on this device whose weight replica (and config) the new instance borrows -- e.g. the same
    decode path with another vocoder precision."""
    from fq3hip.config import qwen3_tts_0p6b, qwen3_tts_1p7b
    from fq3hip.weights import synth_weights
    from fq3hip.model import FasterQwen3TTS
    if share is not None:
        cfg, W = share._bench_cfg, share._bench_weights
    else:
        cfg = qwen3_tts_0p6b() if size == "0p6b" else qwen3_tts_1p7b()
        cfg.tts_model_type = model_type                          # Base / CustomVoice / VoiceDesign share one architecture
        cfg.spk_id = {SPEAKER: cfg.talker.vocab_size - 1024 + 300}
        W = synth_weights(cfg, 0, torch.bfloat16, parts=("talker", "predictor", "codec", "text"), codec_normalized=True)
    model = FasterQwen3TTS.from_weights(cfg, W, device=device, dtype=torch.bfloat16, max_seq_len=max_seq_len,
                                        codec_max_frames=REF_FRAMES + frames + 16, max_frames=frames + 8, share=share,
                                        codec_precision=codec_precision)
    model._bench_weights, model._bench_cfg = W, cfg
    # paged KV: the batch schedulers of this model draw from a pool sized for the traffic of this bench (a ~200-row prompt + `frames`
    # frames per request, 128 lanes + 128 spare contexts) instead of (lanes + spares) x max_seq_len slots: 2048 blocks of 64 keys
    # against 8192 (max_seq_len 2048) or 24576 (6144)
    model.batch_kv_blocks = 256 * ((PROMPT_LEN + frames + 64 + 63) // 64)
    if BATCH_LOOKAHEAD is not None:
        model.batch_lookahead = BATCH_LOOKAHEAD           # --batch-lookahead (measurement switch)
    return cfg, model


def build_request(cfg, device):
    """The synthetic request of SURVEY.md section 8(d) expressed through the PUBLIC API, so that the timed region starts
    where the reference's does (text in, README.md:219 / benchmarks/throughput.py:49-61): an ICL voice-clone prompt with
    170 reference frames, a 20-token instruct turn, 90 + 112 text tokens -> 200 prompt rows (20 + 3 + 6 + 171) and 32
    trailing-text rows.  No HF tokenizer exists offline: the byte-level stand-in of fq3hip.native_model tokenises."""
    g = torch.Generator().manual_seed(1237)
    ref_code = torch.cat([torch.randint(0, cfg.talker.vocab_size - 1024, (REF_FRAMES, 1), generator=g),
                          torch.randint(0, cfg.codec.codebook_size, (REF_FRAMES, cfg.num_code_groups - 1), generator=g)], 1).to(device)
    spk = torch.randn(cfg.talker.hidden_size, generator=g).to(device=device, dtype=torch.bfloat16)
    words = "the quick brown fox jumps over the lazy dog and keeps running through the quiet evening fields "
    return dict(text=(words * 2)[:112], language="English", ref_text=words[:90], instruct="speak calmly!!!",
                voice_clone_prompt=dict(ref_code=[ref_code], ref_spk_embedding=[spk], x_vector_only_mode=[False], icl_mode=[True]))


def prepared_prompt(model, req):
    """(tie, tam, tth, tpe, ref_codes) of the request, for the blocks that time the decode / prefill alone."""
    _m, _talker, _config, tie, tam, tth, tpe, rc = model._prepare_generation(
        text=req["text"], language=req["language"], ref_text=req["ref_text"], voice_clone_prompt=req["voice_clone_prompt"],
        instruct=req["instruct"], non_streaming_mode=False)
    assert tie.shape[1] == PROMPT_LEN and tth.shape[1] == 32 and rc.shape[0] == REF_FRAMES, (tie.shape, tth.shape, rc.shape)
    return [tie, tam, tth, tpe, rc]


def one_utterance(model, req, seed, sync=None, frames=FRAMES):
    """Streaming voice-clone of one synthetic utterance through the public entry point: tokenisation, prompt build,
    prefill, decode, streaming vocoder.  Returns (ttfa_s, wall_s, n_frames, pcm)."""
    sync = sync or torch.cuda.synchronize
    torch.manual_seed(seed)
    sync()
    t0 = time.perf_counter()
    ttfa, chunks, n = None, [], 0
    for audio, sr, timing in model.generate_voice_clone_streaming(
            text=req["text"], language=req["language"], ref_text=req["ref_text"], voice_clone_prompt=req["voice_clone_prompt"],
            instruct=req["instruct"], chunk_size=CHUNK, max_new_tokens=frames, min_new_tokens=frames):
        if ttfa is None:
            ttfa = time.perf_counter() - t0       # `audio` is a host array: the first chunk is complete here
        chunks.append(audio)
        n = timing["total_steps_so_far"]
    sync()
    wall = time.perf_counter() - t0
    return ttfa, wall, n, np.concatenate(chunks) if chunks else np.zeros(0, np.float32)


def measure_frame_graph(model, prompt, n=64, frames=FRAMES):
    """HIP events on the launch stream around n consecutive decode-frame graph replays."""
    from fq3hip.generate import _prefill_and_arm, run_frames
    tie, tam, tth, tpe, _ = prompt
    m = model.model.model
    eng, tn, pn, _ = _prefill_and_arm(m.talker, tie, tam, tth, tpe, m.config.talker_config, model.predictor_graph,
                                      model.talker_graph, frames, frames, 0.9, 50, 1.0, True, 1.05, use_graph=True)
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


def frame_roofline(cfg, frame_ms, p_mid, lanes=1, what="decode-frame hipGraph"):
    b = algorithmic_bytes_per_frame(cfg, p_mid, lanes)
    ach = b / (frame_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": what, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(b), "kv_len": p_mid, "launch_ms": round(frame_ms, 4)}
    if lanes > 1:
        # the weights are read ONCE per lock-step frame, so the byte count above barely grows with the lanes; what the
        # batch replaces is `lanes` single-stream frames, each of which would stream the weights itself:
        eq = lanes * algorithmic_bytes_per_frame(cfg, p_mid, 1) / (frame_ms * 1e-3) / 1e9
        out["unbatched_equivalent"] = {"bandwidth_needed": round(eq, 1), "unit": "GB/s", "times_hbm_peak": round(eq / HBM_PEAK_GBS, 2),
                                       "note": f"NOT a roofline fraction: the bandwidth {lanes} independent single-stream decodes would need for the same "
                                               "frame rate (each streaming the weights itself); above 1.0 x peak means unbatched decoding could not reach it"}
    return out


def measure_mfma(cfg, model, prompt):
    """Event-timed matrix-core work of the path: one full-length codec decode (ref + generated frames, the non-streaming
    call) and one 200-token prefill.  FLOPs are the GEMM FLOPs of the shapes (attention / elementwise excluded)."""
    tok = model.model.model.speech_tokenizer
    T = REF_FRAMES + FRAMES
    g = torch.Generator().manual_seed(4)
    codes = torch.randint(0, cfg.codec.codebook_size, (T, cfg.codec.num_quantizers), generator=g).to(tok.device)
    tok.decode_tensor(codes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        tok.decode_tensor(codes)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    fl = sum(codec_gemm_flops(cfg.codec, e - s + c) for s, e, c in tok._pieces(T))
    out = {"codec_full_decode": {"frames": T, "ms": round(ms, 3), "gemm_tflop": round(fl / 1e12, 3),
                                 "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "bound": "mfma"}}
    eng = model.talker_graph.engine
    x = prompt[0][0].contiguous()
    eng.prefill(x)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        eng.prefill(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = prefill_gemm_flops(cfg, x.shape[0])
    out["prefill_200"] = {"tokens": int(x.shape[0]), "ms": round(ms, 3), "gemm_tflop": round(fl / 1e12, 3),
                          "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "bound": "mfma"}
    # the PACKED prefill of the batch scheduler's staged admission (fq3_prefill_batch: the rows of several prompts through one pass over
    # the weights; a pack holds as many prompts as fit the leading context's max_seq_len rows -- 10 x 200 at 2048)
    try:
        from fq3hip.engine import Fq3Engine, Fq3KvPool
        n = max(1, min(10, int(eng.max_seq_len) // int(x.shape[0])))
        device = eng.device
        pool = Fq3KvPool(eng.cfg, n * 5, device=device, dtype=eng.dtype)
        engs = [Fq3Engine(eng.cfg, eng.weights, device=device, dtype=eng.dtype, max_seq_len=eng.max_seq_len, max_frames=8, share=eng, pool=pool)
                for _ in range(n)]
        engs[0].prefill_reserve()
        xs = [x] * n
        for _ in range(3):
            Fq3Engine.prefill_batch(engs, xs)
        torch.cuda.synchronize(device)
        e0.record()
        for _ in range(10):
            Fq3Engine.prefill_batch(engs, xs)
        e1.record(); torch.cuda.synchronize(device)
        msp = e0.elapsed_time(e1) / 10
        out[f"packed_prefill_{n}x{int(x.shape[0])}"] = {"prompts": n, "tokens_each": int(x.shape[0]), "ms": round(msp, 3), "ms_per_prompt": round(msp / n, 3),
                                                         "gemm_tflop": round(n * fl / 1e12, 3), "achieved": round(n * fl / (msp * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                                                         "unit": "TFLOP/s", "frac": round(n * fl / (msp * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "bound": "mfma"}
        for e in engs:
            e.close()
    except Exception as exc:                  # a measurement row, never a reason to lose the line
        out["packed_prefill"] = {"error": repr(exc)}
    return out


def concurrent_throughput(cfg, model, req, device, streams=4, utterances=2):
    """Extra (not the headline): S utterances in flight on ONE GPU, each with its own decode context, codec
    workspace, hipGraph and HIP stream, all borrowing the single weight replica."""
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
                one_utterance(models[i], req, 5000 + i, sync=st.synchronize)     # per-context warm-up + graph capture
                bar.wait(timeout=60)
                t0 = time.perf_counter()
                frames, ttfas = 0, []
                for u in range(utterances):
                    ttfa, wall, n, _ = one_utterance(models[i], req, 6000 + 10 * i + u, sync=st.synchronize)
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


def batched_run(model, prompt, n_utt, lanes=8, frames=FRAMES, seed0=2000, sync=None):
    """`n_utt` synthetic utterances (seeds seed0 + i) through `lanes` lock-step lanes (fq3_batch_*, continuous batching)
    and the vocoder, exactly as generate_voice_clone_batch runs them behind its prompt builder (FasterQwen3TTS._run_batch_full:
    staged admission, waveforms produced in slices on the side stream while the utterances decode, utterances at the same point
    vocoded as one batch).  Returns (audio_s, wall_s, pcm lengths)."""
    tie, tam, tth, tpe, ref_codes = prompt
    m = model.model.model
    talker, config = m.talker, m.config.talker_config
    kw = model._gen_kwargs(frames, frames, 0.9, 50, 1.0, True, 1.05)
    sync = sync or torch.cuda.synchronize
    torch.manual_seed(seed0)
    sync()
    t0 = time.perf_counter()
    outs = model._run_batch_full([(talker, config, tie, tam, tth, tpe, ref_codes)] * n_utt, kw, lanes, n_utt)
    sync()
    wall = time.perf_counter() - t0
    lens = [int(len(w[0])) for w, _sr in outs]
    want = model.model.model.speech_tokenizer.num_samples_total(ref_codes.shape[0] + frames)
    want -= int(ref_codes.shape[0] / (ref_codes.shape[0] + frames) * want)
    assert all(n == want for n in lens), (sorted(set(lens)), want)          # every utterance ran its `frames` frames (EOS suppressed)
    return n_utt * frames * FRAME_S, wall, lens


def batched_streaming_run(model, req, lanes=8, n_utt=16):
    """Streaming AND batched through the public entry point (generate_voice_clone_batch_streaming): `n_utt` requests handed
    over at once, `lanes` lock-step lanes, every utterance vocoded chunk by chunk on the side stream.  TTFA = time from the
    call to an utterance's first audio chunk on the host (first wave = the requests that get a lane immediately)."""
    torch.manual_seed(4242)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first, samples = {}, 0
    for i, audio, sr, tm in model.generate_voice_clone_batch_streaming(
            [req["text"]] * n_utt, language=req["language"], ref_text=req["ref_text"], voice_clone_prompt=req["voice_clone_prompt"],
            instruct=req["instruct"], chunk_size=CHUNK, max_new_tokens=FRAMES, min_new_tokens=FRAMES, lanes=lanes):
        first.setdefault(i, time.perf_counter() - t0)
        samples += len(audio)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    wave = sorted(first[i] for i in range(min(lanes, n_utt)))
    return {"utterances": n_utt, "lanes": lanes, "value": round(samples / 24000.0 / wall, 3),
            "unit": "x real-time (aggregate audio s / wall s; tokenise + prompt build + staged prefill + lock-step decode + streaming vocoder)",
            "ttfa_ms_first_wave_p50": round(1000 * float(np.median(wave)), 2), "ttfa_ms_first_wave_max": round(1000 * wave[-1], 2)}


def batched_groups_run(cfg, model, prompt, device, n_utt, groups=2, lanes=8):
    """Batching AND concurrency: `groups` independent lock-step batches of `lanes` lanes each (own contexts, own hipGraph,
    own HIP stream and host thread; ONE weight replica) decode at the same time.  A lock-step frame is still a chain of
    latency-bound launches that leaves most CUs idle, so a second batch overlaps it almost for free."""
    import threading
    from fq3hip.model import FasterQwen3TTS
    models = [model] + [FasterQwen3TTS.from_weights(cfg, model._bench_weights, device=device, dtype=torch.bfloat16, max_seq_len=2048,
                                                    codec_max_frames=REF_FRAMES + FRAMES + 16, max_frames=FRAMES + 8, share=model)
                        for _ in range(groups - 1)]
    shares = [list(range(g, n_utt, groups)) for g in range(groups)]
    bar = threading.Barrier(groups)
    res, errors = [None] * groups, []

    def worker(g):
        try:
            torch.cuda.set_device(torch.device(device))
            st = torch.cuda.Stream(device=device)
            with torch.cuda.stream(st):
                batched_run(models[g], prompt, min(lanes, len(shares[g])), lanes, sync=st.synchronize)          # contexts + graph capture
                bar.wait(timeout=120)
                t0 = time.perf_counter()
                audio_s, _w, _l = batched_run(models[g], prompt, len(shares[g]), lanes, seed0=3000 + g, sync=st.synchronize)
                res[g] = (t0, time.perf_counter(), audio_s)
        except BaseException as e:
            errors.append(repr(e))
            bar.abort()

    th = [threading.Thread(target=worker, args=(g,), daemon=True) for g in range(groups)]
    for t in th:
        t.start()
    deadline = time.time() + 240
    for t in th:
        t.join(timeout=max(1.0, deadline - time.time()))
    if errors or any(r is None for r in res):
        return {"groups": groups, "error": "; ".join(errors) or "worker timed out"}
    t0 = min(r[0] for r in res); t1 = max(r[1] for r in res)
    return {"groups": groups, "lanes_per_group": lanes, "utterances": n_utt, "value": round(sum(r[2] for r in res) / (t1 - t0), 3),
            "unit": "x real-time (aggregate audio s / wall s, one GPU, prefill + lock-step decode + non-streaming vocoder)",
            "wall_s": round(t1 - t0, 3)}


def batched_frame_time(model, cfg, prompt, lanes=8, n=48, mfma=1):
    """HIP events around n lock-step frame graph replays with all lanes armed (decode only)."""
    from fq3hip.generate import _prefill_and_arm, NOISE_RING
    tie, tam, tth, tpe, _ = prompt
    m = model.model.model
    dec = model._batch_decoder(lanes)
    dec.batch.set_option("mfma", int(mfma))
    keep = []
    for ln in dec.lanes:
        keep.append(_prefill_and_arm(m.talker, tie, tam, tth, tpe, m.config.talker_config, ln.predictor_graph, ln.talker_graph,
                                     FRAMES, FRAMES, 0.9, 50, 1.0, True, 1.05, use_graph=False))
    for _e, tn, pn, _ in keep:
        tn.exponential_(1); pn.exponential_(1)
    dec.batch.graph_capture()
    dec._captured = True
    dec.batch.frames(8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dec.batch.frames(n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    # let every lane run out so the decoder's lanes are reusable
    dec.batch.frames(NOISE_RING - 8 - n)
    torch.cuda.synchronize()
    dec.batch.set_option("mfma", 1)
    dec._captured = False
    return ms, PROMPT_LEN + 8 + n / 2


def parity_note(cfg, model):
    """Teacher-forced agreement of THIS model (0.6B shapes, bf16, full depth, hipGraph replay) with the CPU oracle's
    golden ids (tests/golden/fulldepth.npz; same seeded weights and prompt).  oracle/ is used as the checker only."""
    from oracle import teacher_forced as TF
    from fq3hip.weights import synth_prompt
    g = np.load(os.path.join(ROOT, "tests", "golden", "fulldepth.npz"))
    frames, plen, tlen = (int(x) for x in g["meta"])
    case = TF.load_case(g, "0p6b_bf16")
    tie, tam, tth, tpe, _ = synth_prompt(cfg, plen, tlen, 0, dtype=torch.bfloat16)
    eng = model.talker_graph.engine
    pg = model.predictor_graph
    saved = dict(do_sample=pg.do_sample, top_k=pg.top_k, temperature=pg.temperature)
    pg.do_sample, pg.top_k, pg.temperature = False, 0, 1.0            # greedy predictor, as the golden ids were made
    try:
        dec = TF.forced_decisions(eng, cfg, tie, tth, tpe, case["codes"], graph=True)
    finally:
        pg.do_sample, pg.top_k, pg.temperature = saved["do_sample"], saved["top_k"], saved["temperature"]
    s = TF.score(dec, case, 3.0)
    return {"matched_frames": s["matched_frames"], "frames": s["frames"], "matched_decisions": s["matched_decisions"],
            "decisions": s["total"], "worst_mismatch_margin_bf16_ulp": s["worst_mismatch_ulp"], "unexplained": s["unexplained"],
            "method": "teacher-forced vs CPU-oracle golden ids, 28+5 layers, 200-token prompt; a mismatch counts as explained "
                      "when the oracle's own top-2 margin is <= 3 bf16 ulps (tests/test_gpu_fulldepth.py)"}


def parity_pcm(cfg, model_bf16, model, device):
    """PCM parity of THIS model's codec weights (bf16 checkpoint values) on a bounded sample: the committed golden waveform of
    the fp32-arithmetic CPU oracle on the same bf16-valued weights (tests/golden/codec_real_q.npz, T = 100 frames > the
    attention window; oracle/make_golden_codec_real.py) against (a) the vocoder every figure of this line ran with
    (HEADLINE_CODEC), (b) the bf16 vocoder (what the reference's Torch path runs), (c) the all-fp32 mode, with the cost of each for
    the full 370-frame decode, a streaming chunk, and -- batched -- 16 utterances in one launch set."""
    from fq3hip.codec import HipSpeechTokenizer
    g = np.load(os.path.join(ROOT, "tests", "golden", "codec_real_q.npz"))
    T = 100
    codes = torch.from_numpy(g[f"codes_{T}"].astype(np.int64)).to(device)
    ref = g[f"pcm_f32q_{T}"]
    rms = lambda a: float(np.sqrt(np.mean(np.square(a.astype(np.float64)))))
    toks = {HEADLINE_CODEC: model.model.model.speech_tokenizer, "bf16": model_bf16.model.model.speech_tokenizer}
    own = None
    if "fp32" not in toks:
        own = toks["fp32"] = HipSpeechTokenizer(cfg.codec, model._bench_weights, str(device), max_frames=REF_FRAMES + FRAMES + 16, precision="fp32")
    gg = torch.Generator().manual_seed(4)
    full = torch.randint(0, cfg.codec.codebook_size, (REF_FRAMES + FRAMES, cfg.codec.num_quantizers), generator=gg).to(device)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {"sample": f"T = {T} frames ({ref.size} samples), fp32-arithmetic CPU-oracle golden on the bf16 checkpoint weights", "signal_rms": round(rms(ref), 4),
           "tolerance_north_star": 1e-3, "vocoder_of_this_line": HEADLINE_CODEC,
           "reference": "fp32 arithmetic on the bf16 checkpoint weights (tests/golden/codec_real_q.npz) -- NOT the reference's own bf16 Torch arithmetic: "
                        "the CPU oracle's bf16 evaluation of this synthetic vocoder is itself 7.6e-3 RMS from its fp32 evaluation, so no "
                        "implementation can be within 1e-3 of both; distances to the bf16-arithmetic oracle are in *_vs_bf16_oracle"}
    gb = np.load(os.path.join(ROOT, "tests", "golden", "codec_real.npz"))
    ref_b16 = None
    if f"pcm_bf16bits_{T}" in gb and np.array_equal(gb[f"codes_{T}"], g[f"codes_{T}"]):
        ref_b16 = torch.from_numpy(gb[f"pcm_bf16bits_{T}"]).view(torch.bfloat16).float().numpy()      # the CPU oracle's bf16-arithmetic waveform, same codes and weights
        out["bf16_oracle_vs_fp32_oracle"] = float(f"{rms(ref_b16 - ref):.3e}")
    for name, tok in toks.items():
        wav = tok.decode_tensor(codes).cpu().numpy()
        w = full[:33].contiguous()
        first = tok.num_samples_total(33) - 8 * 1920
        row = {"pcm_rms_vs_fp32_oracle": float(f"{rms(wav - ref):.3e}"),
               "pcm_rms_vs_bf16_oracle": float(f"{rms(wav - ref_b16):.3e}") if ref_b16 is not None and ref_b16.shape == wav.shape else None,
               "full_decode_370_frames_ms": round(timed(lambda: tok.decode_tensor(full)), 3),
               "streaming_chunk_8_frames_ms": round(timed(lambda: tok.decode_tensor(w, first), 5), 3)}
        if name != "fp32":
            b16 = full.unsqueeze(0).repeat(16, 1, 1).contiguous()
            cut = int(REF_FRAMES / (REF_FRAMES + FRAMES) * tok.num_samples_total(REF_FRAMES + FRAMES))
            # as the product's batch path decodes them (fq3hip/model.py::_SideVocoder.submit_many): behind the voice's cached reference-prefix
            # state (bit-identical to the full front end, tests/test_gpu_codec.py); the figure without it is kept beside it
            pf = tok.prefix_for(full[:REF_FRAMES].contiguous())
            row["batched_16_utterances_tail_after_reference_ms_per_utterance"] = round(timed(lambda: tok.decode_tensor_batch(b16, cut, prefixes=[pf] * 16), 2) / 16, 3)
            row["batched_16_utterances_tail_after_reference_ms_per_utterance_without_prefix_state"] = round(timed(lambda: tok.decode_tensor_batch(b16, cut), 2) / 16, 3)
            wf = full[:REF_FRAMES + 8].contiguous()
            f8 = tok.num_samples_total(REF_FRAMES + 8) - 8 * 1920
            row["first_streaming_chunk_behind_prefix_state_ms"] = round(timed(lambda: tok.decode_tensor(wf, f8, prefix=pf), 5), 3)
            b32 = wf.unsqueeze(0).repeat(32, 1, 1).contiguous()
            row["first_streaming_chunks_32_batched_ms_each"] = round(timed(lambda: tok.decode_tensor_batch(b32, f8, prefixes=[pf] * 32), 3) / 32, 3)
        row["meets_1e-3"] = bool(row["pcm_rms_vs_fp32_oracle"] <= 1e-3)
        out[name] = row
    out["note"] = ("bf16 arithmetic of this synthetic vocoder is chaotic at the 7.6e-3 level (the CPU oracle's own bf16 run vs its fp32 run on "
                   "the same weights); bf16x2 keeps bf16 weights and carries every activation as bf16 high part + residual (two bf16 MFMAs per "
                   "product), fp32 widens everything")
    if own is not None:
        own.close()
    return out


def frame_rows(per_dispatch, first_name):
    """[(kernel name, order key, counter value)] in dispatch order -> [(kernel name, dispatches, summed value)] over the dispatches
    from the first one whose name contains `first_name` on (a frame's first kernel): whatever ran before it -- model set-up, the
    lanes' prefills -- is not frame traffic even where it shares kernels with the frame."""
    start = next((i for i, (name, _k, _v) in enumerate(per_dispatch) if first_name in name), None)
    if start is None:
        return []
    acc = {}
    for name, _k, v in per_dispatch[start:]:
        n, s = acc.get(name, (0, 0.0))
        acc[name] = (n + 1, s + float(v))
    return [(name, n, s) for name, (n, s) in acc.items()]


def measure_frame_traffic(timeout_s=300, lanes=0):
    """`lanes` > 0: the lock-step frame of that many lanes (tools/pmc_workload.py batch <lanes>) instead of the single-stream one.
    HBM-side traffic of one decode frame, measured in THIS run: a separate `rocprofv3 --pmc FETCH_SIZE` pass (counters only,
    no trace domains: MI355X_MICROARCH.md, HBM section) over tools/pmc_workload.py (the same model, direct launches of the frame's
    kernels), summed over the frame's kernels and divided by the number of frames that ran (one frame_begin_kernel dispatch each).
    FETCH_SIZE is reported in KB and counts 64 B per 128-B fabric request on gfx950: x2.  Returns bytes per frame, or None."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="fq3_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [exe, "--pmc", "FETCH_SIZE", "-d", out, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py")] + \
          (["batch", str(int(lanes))] if lanes > 0 else ["frames"])
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
        dbs = [os.path.join(dp, f) for dp, _dn, fs in os.walk(out) for f in fs if f.endswith(".db")]
        if not dbs:
            return None, "the counter pass produced no database"
        cur = sqlite3.connect(dbs[0]).cursor()
        # one row per dispatch, in dispatch order: the lanes' PREFILLS run through some of the frame's kernels too (the
        # weight-stationary GEMM serves both), so only dispatches from the first frame's first kernel on are frame traffic
        dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
        order = next((c for c in ("dispatch_id", "start", "id") if c in dcols), "event_id")
        per = cur.execute(f"""select s.kernel_name, d.{order}, e.value from rocpd_pmc_event e
                              join rocpd_info_pmc p on e.pmc_id = p.id
                              join rocpd_kernel_dispatch d on e.event_id = d.event_id
                              join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                              where p.name = 'FETCH_SIZE' order by d.{order}""").fetchall()
        first_name = "frame_begin_batch_kernel" if lanes > 0 else "frame_begin_kernel"
        rows = frame_rows(per, first_name)
        # every dispatch from the first frame's first kernel on IS frame traffic (tools/pmc_workload.py runs nothing but frames after it);
        # a name filter here once missed a kernel the frame had gained (the lane attention: 3.7 of 9.9 GB per 128-lane frame)
        kb = sum(v for _name, _n, v in rows)
        frames = sum(n for name, n, _v in rows if first_name in name)
        if frames <= 0 or kb <= 0:
            return None, "no decode-frame dispatches in the counter pass"
        return 2.0 * 1024.0 * kb / frames, f"{int(frames)} profiled frames"
    except Exception as e:
        return None, repr(e)
    finally:
        shutil.rmtree(out, ignore_errors=True)


def cpu_baseline(cfg, frames=40, budget_s=45.0):
    """CPU oracle ("port") on the host cores: same shapes, fp32 (bf16 matmuls are not accelerated on every host),
    same prompt; a BOUNDED sample: 200-token prefill + up to `frames` generated frames + their vocoding, cut short when
    `budget_s` is exceeded.  RTF with the same definition as the GPU line."""
    from fq3hip.weights import synth_weights, synth_prompt
    from oracle import qwen3tts_oracle as O
    cores = min(os.cpu_count() or 1, 16)        # batch-1 matvecs stop scaling (and regress) beyond ~16 threads
    torch.set_num_threads(cores)
    W = synth_weights(cfg, 0, torch.float32, parts=("talker", "predictor", "codec"), codec_normalized=True)
    tie, tam, tth, tpe, ref = synth_prompt(cfg, PROMPT_LEN, 32, REF_FRAMES, dtype=torch.float32)
    orc = O.OracleTTS(cfg, W, max_seq_len=512)
    t0 = time.perf_counter()
    with torch.inference_mode():
        # frame-by-frame so the budget can stop the sample: first a 1-frame run (prefill + frame), then more
        sp = O.SamplingParams(max_new_tokens=1, min_new_tokens=1)
        codes = orc.generate(tie, tam, tth, tpe, sp)
        t1 = time.perf_counter() - t0
        if t1 < budget_s / 3:
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
        O.codec_decode(codes % cfg.codec.codebook_size, W, cfg.codec)
        t_voc = time.perf_counter() - tv
    n = codes.shape[0]
    wall = t_codes + t_voc
    return {"value": round(n * FRAME_S / wall, 5), "unit": "x real-time (audio s / wall s)", "cores": cores, "kind": "port",
            "sample": f"oracle/qwen3tts_oracle.py fp32 eager Torch, 0.6B shapes, {PROMPT_LEN}-token prefill + {n} frames "
                      f"+ vocoding of those frames ({t_codes:.1f}s + {t_voc:.1f}s); host has {os.cpu_count()} cores, {cores} threads used",
            "ms_per_frame": round(1000 * t_codes / n, 1)}


def reference_wave(seconds: float, seed: int = 31) -> np.ndarray:
    """A speech-like synthetic clip (harmonics with a slow envelope + noise), 24 kHz mono float32."""
    n = int(24000 * seconds)
    t = np.arange(n, dtype=np.float64) / 24000.0
    rng = np.random.default_rng(seed)
    x = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in ((0.2, 180.0, 0.0), (0.12, 360.0, 0.5), (0.08, 1240.0, 1.1), (0.04, 3100.0, 2.0)))
    x = x * (0.55 + 0.45 * np.sin(2 * np.pi * 2.5 * t)) + 0.02 * rng.standard_normal(n)
    return x.astype(np.float32)


def ref_analysis_block(cfg, device, seconds=10.0, reps=10):
    """SURVEY section 8f row 1, second half: what the reference does once per new reference clip (model.py:430-447,
    upstream create_voice_clone_prompt) -- speech-tokenizer encoder + speaker encoder -- on the HIP analysers, for a
    `seconds` clip + the 0.5 s of appended silence; and the same arithmetic through the CPU oracle (fp32 Torch)."""
    from fq3hip.refenc import HipRefAudioAnalyzer
    from fq3hip.weights import synth_ref_audio_weights
    rc = cfg.ref_audio
    W = synth_ref_audio_weights(rc, 0)
    an = HipRefAudioAnalyzer(rc, W, device=str(device))
    wav = np.concatenate([reference_wave(seconds), np.zeros(12000, np.float32)])
    x = torch.from_numpy(wav).to(device)
    an.encode(x); an.speaker_embedding(x)
    torch.cuda.synchronize()

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    enc_ms, spk_ms = timed(lambda: an.encode(x)), timed(lambda: an.speaker_embedding(x))
    t0 = time.perf_counter()
    for _ in range(reps):
        codes = an.encode(wav); emb = an.speaker_embedding(wav)        # host buffer in: PCIe-inclusive
        torch.cuda.synchronize()
    host_ms = 1000 * (time.perf_counter() - t0) / reps
    out = {"clip_s": seconds + 0.5, "frames": int(codes.shape[0]), "tokenizer_encoder_ms": round(enc_ms, 3), "speaker_encoder_ms": round(spk_ms, 3),
           "both_from_host_buffer_ms": round(host_ms, 3), "dtype": "f32",
           "what": "fq3_refenc_encode + fq3_refenc_speaker (Mimi-shape encoder: 64..1024-channel SEANet, 8 x 512-d layers, 16 x 2048 RVQ; "
                   "128-bin log-mel + ECAPA-TDNN 512/1536 channels); synthetic seeded weights"}
    try:
        from oracle import refenc_oracle as RO                            # CPU baseline leg of this block
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        xc = torch.from_numpy(wav)
        t0 = time.perf_counter()
        with torch.inference_mode():
            ref_codes, margins = RO.tokenizer_encode(W, rc, xc, return_all=True)[:2]
            ref_emb, _ = RO.speaker_embedding(W, rc, xc)
        out["cpu_oracle_ms"] = round(1000 * (time.perf_counter() - t0), 1)
        same = (codes.cpu() == ref_codes)
        out["parity"] = {"ids_identical": int(same.sum()), "ids": int(same.numel()),
                         "first_level_identical": int(same[:, 0].sum()), "frames": int(same.shape[0]),
                         "speaker_embedding_max_abs_diff": float((emb.cpu() - ref_emb).abs().max()),
                         "note": "ids after a near-tie arg-min of a frame's residual chain may differ (tests/test_gpu_refenc.py attributes them)"}
    except Exception as e:
        out["cpu_oracle_ms"] = {"error": repr(e)}
    return out


def sentences(n: int):
    """`n` distinct ~110-character lines (the utterances of the configs[3] block)."""
    words = ("the quick brown fox jumps over the lazy dog and keeps running through the quiet evening fields while "
             "a small boat drifts along the river past the old stone bridge under a pale and patient moon ").split()
    out = []
    for i in range(n):
        k = (7 * i) % len(words)
        line = " ".join(words[k:] + words[:k])
        out.append((f"line {i}: " + line)[:110])
    return out


def custom_voice_batch_run(model, texts, lanes, frames=FRAMES, seed=2000):
    """BASELINE configs[3] on this rank's share: CustomVoice utterances (speaker-id prompts, model.py:1139-1326) through the
    public generate_custom_voice_batch -- text in, waveforms on the host out.  Returns (audio_s, wall_s, sample counts)."""
    torch.manual_seed(seed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = model.generate_custom_voice_batch(texts, SPEAKER, "English", max_new_tokens=frames, min_new_tokens=frames, lanes=lanes)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    lens = [int(len(w[0])) for w, _sr in outs]
    return sum(lens) / 24000.0, wall, lens


def config4_block(cfg, model, device, prompt_rows=4096, runs=2):
    """BASELINE configs[4]: VoiceDesign long-form -- a `prompt_rows`-row instruct + text prompt (non_streaming_mode = the
    VoiceDesign default, model.py:1348-1351: the whole text sits in the prompt), max_seq_len >= prompt + frames, streamed in
    8-frame chunks through the public generate_voice_design_streaming: device prompt build (text projection of ~4k tokens),
    matrix-core prefill (flash attention + 256-wide GEMM tiles), 200 decode frames at KV 4096..4296, chunked codec decode."""
    inner = model.model.model
    saved = inner.tts_model_type
    inner.tts_model_type = "voice_design"
    try:
        text = sentences(1)[0]
        base = "a calm, low and steady narrator who speaks slowly, with long pauses and a warm tone; "
        instruct = (base * (prompt_rows // len(base) + 2))[:prompt_rows - 200]
        for _ in range(4):                      # byte-level stand-in tokenizer: trim / extend the instruct to hit the row count exactly
            _m, _t, _c, tie, _tam, _tth, _tpe = model._design_prepare(text, instruct, "English", None)
            d = prompt_rows - int(tie.shape[1])
            if d == 0:
                break
            instruct = instruct + base[:d] if d > 0 else instruct[:d]
        rows = int(tie.shape[1])

        def one(seed):
            torch.manual_seed(seed)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ttfa, n, samples = None, 0, 0
            for audio, _sr, tm in model.generate_voice_design_streaming(text, instruct, "English", chunk_size=CHUNK,
                                                                        max_new_tokens=FRAMES, min_new_tokens=FRAMES):
                if ttfa is None:
                    ttfa = time.perf_counter() - t0
                n = tm["total_steps_so_far"]
                samples += len(audio)
            torch.cuda.synchronize()
            return ttfa, time.perf_counter() - t0, n, samples

        one(700)
        res = [one(701 + i) for i in range(runs)]
        # the decode-frame graph at this context length (KV ~ 4096 + 100)
        return {"workload": f"configs[4]: Qwen3-TTS-12Hz-1.7B-VoiceDesign shapes, {rows}-row prompt ({len(instruct)} instruct bytes + text, "
                            f"byte-level stand-in tokenizer), streaming chunk_size={CHUNK}, {FRAMES} frames, max_seq_len {model.max_seq_len}",
                "prompt_rows": rows, "rtf": round(float(np.mean([n * FRAME_S / w for _t, w, n, _s in res])), 3),
                "ttfa_ms_p50": round(1000 * float(np.median([t for t, _w, _n, _s in res])), 2),
                "wall_ms_per_utterance": round(1000 * float(np.mean([w for _t, w, _n, _s in res])), 1),
                "frames": int(res[0][2]), "samples": int(res[0][3])}
    finally:
        inner.tts_model_type = saved


def model_1p7b_block(cfg, model, device, lanes=16):
    """BASELINE configs[2]: the 1.7B shapes (talker hidden 2048 / intermediate 6144, predictor with projection), single
    stream: RTF / TTFA over 2 utterances + the decode-frame roofline; the lock-step batch at these shapes; the 4096-token prefill;
    and BASELINE configs[4] end to end (config4_voice_design_4k).  The single-stream figures (rtf / ttfa_ms_p50 here and in
    config4_voice_design_4k) run with the 1e-3-compliant vocoder (HEADLINE_CODEC, `model`'s); `*_bf16_codec` are the same runs with a
    bf16-vocoder sibling instance over the same weight replica."""
    req = build_request(cfg, device)
    _c, model_lo = build_model(device, max_seq_len=model.max_seq_len, share=model)          # bf16 vocoder over the same weight replica
    model_lo.model.model.tts_model_type = model.model.model.tts_model_type
    one_utterance(model, req, 900)
    for i in range(2):
        one_utterance(model_lo, req, 901 + i)
    prompt = prepared_prompt(model, req)
    frame_ms, p_mid = measure_frame_graph(model, prompt)
    res = [one_utterance(model, req, 910 + i) for i in range(3)]
    res_lo = [one_utterance(model_lo, req, 910 + i) for i in range(3)]
    out = {"workload": "configs[2]: Qwen3-TTS-12Hz-1.7B-Base shapes, voice-clone streaming chunk_size=8, bf16, synthetic weights",
           "vocoder": f"codec_precision={HEADLINE_CODEC!r} (meets the 1e-3 PCM bound, parity_pcm); *_bf16_codec: the bf16 vocoder",
           "rtf": round(float(np.mean([n * FRAME_S / w for _, w, n, _ in res])), 3),
           "ttfa_ms_p50": round(1000 * float(np.median([t for t, _, _, _ in res])), 2),
           "rtf_bf16_codec": round(float(np.mean([n * FRAME_S / w for _, w, n, _ in res_lo])), 3),
           "ttfa_ms_p50_bf16_codec": round(1000 * float(np.median([t for t, _, _, _ in res_lo])), 2),
           "decode_ms_per_frame": round(frame_ms, 4), "roofline": frame_roofline(cfg, frame_ms, p_mid)}
    try:
        # BASELINE configs[4] shape: a 4096-token prompt (synthetic embeddings) through the matrix-core prefill
        from fq3hip.weights import synth_prompt
        eng = model.talker_graph.engine
        x = (synth_prompt(cfg, 4096, 4, 0, dtype=torch.bfloat16)[0][0] * 30).to(torch.bfloat16).to(device).contiguous()
        eng.prefill(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.prefill(x)
        e1.record()
        torch.cuda.synchronize()
        pms = e0.elapsed_time(e1)
        t = cfg.talker
        gemm_fl = prefill_gemm_flops(cfg, 4096)
        attn_fl = 4.0 * (4096 * 4097 / 2) * t.head_dim * t.num_attention_heads * t.num_hidden_layers
        out["prefill_4096"] = {"workload": "configs[4] shape: 4096-token prompt, 28 layers at 1.7B dims, flash MFMA attention + 256x256 ring / glds GEMMs",
                               "ms": round(pms, 2), "gemm_tflop": round(gemm_fl / 1e12, 2), "attention_tflop": round(attn_fl / 1e12, 2),
                               "achieved": round((gemm_fl + attn_fl) / (pms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round((gemm_fl + attn_fl) / (pms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "bound": "mfma"}
    except Exception as e:
        out["prefill_4096"] = {"error": repr(e)}
    for B in sorted({16, 32, 64, lanes}):
        try:
            ms, p = batched_frame_time(model, cfg, prompt, lanes=B)
            out[f"batched_b{B}"] = {"ms_per_lockstep_frame": round(ms, 3), "value": round(B * 80.0 / ms, 1),
                                    "unit": f"x real-time aggregate, decode only ({B} lanes, one GPU)",
                                    "roofline": frame_roofline(cfg, ms, p, lanes=B, what=f"batched decode-frame hipGraph, {B} lanes")}
        except Exception as e:
            out[f"batched_b{B}"] = {"error": repr(e)}
    try:
        c4 = config4_block(cfg, model, device)
        lo = config4_block(cfg, model_lo, device, runs=2)
        c4["vocoder"] = f"codec_precision={HEADLINE_CODEC!r}; *_bf16_codec: the bf16 vocoder"
        c4["rtf_bf16_codec"], c4["ttfa_ms_p50_bf16_codec"] = lo["rtf"], lo["ttfa_ms_p50"]
        c4["parity"] = config4_parity(cfg, model)
        out["config4_voice_design_4k"] = c4
    except Exception as e:
        out["config4_voice_design_4k"] = {"error": repr(e)}
    return out


def config4_parity(cfg, model):
    """Parity of the configs[4] shape AT FULL DEPTH on this very model (1.7B shapes, bf16, 28 + 5 layers): the 4096-token prefill's
    last-position hidden state / logits and 8 teacher-forced frames over the > 4096-key cache against the CPU oracle's goldens
    (tests/golden/longprompt_full.npz, oracle/make_golden_longprompt_full.py; same seeded weights and prompt).  oracle/ is the
    checker only (tests/test_gpu_longprompt.py holds the gates)."""
    from oracle import teacher_forced as TF
    from fq3hip.weights import synth_prompt
    g = np.load(os.path.join(ROOT, "tests", "golden", "longprompt_full.npz"))
    frames, plen, tlen = (int(x) for x in g["meta"])
    case = TF.load_case(g, "1p7b_bf16")
    tie, tam, tth, tpe, _ = synth_prompt(cfg, plen, tlen, 0, dtype=torch.bfloat16)
    eng = model.talker_graph.engine
    if eng.max_seq_len < plen + frames + 1:
        return {"skipped": f"max_seq_len {eng.max_seq_len} < {plen + frames + 1}"}
    pg = model.predictor_graph
    saved = (pg.do_sample, pg.top_k, pg.temperature)
    pg.do_sample, pg.top_k, pg.temperature = False, 0, 1.0
    try:
        logits, hidden = eng.prefill(tie[0].to(eng.device).contiguous())
        lg, hd = logits.float().cpu().numpy(), hidden.float().cpu().numpy()
        dec = TF.forced_decisions(eng, cfg, tie, tth, tpe, case["codes"], graph=True)
    finally:
        pg.do_sample, pg.top_k, pg.temperature = saved
    sc = TF.score(dec, case, 4.0)
    ref_l, ref_h = g["1p7b_bf16_logits"], g["1p7b_bf16_hidden"]
    return {"prefill_hidden_max_err_over_scale": float(f"{np.abs(hd - ref_h).max() / max(1.0, np.abs(ref_h).max()):.3e}"),
            "prefill_logits_max_err_over_scale": float(f"{np.abs(lg - ref_l).max() / max(1.0, np.abs(ref_l).max()):.3e}"),
            "matched_decisions": sc["matched_decisions"], "decisions": sc["total"], "matched_frames": sc["matched_frames"],
            "frames": sc["frames"], "worst_mismatch_margin_bf16_ulp": sc["worst_mismatch_ulp"], "unexplained": sc["unexplained"],
            "method": "28 + 5 layers, 4096-token prompt, teacher-forced vs CPU-oracle golden ids; a mismatch counts as explained when the oracle's own top-2 "
                      "margin is <= 4 bf16 ulps -- the floor the oracle's own re-evaluation shows at this shape (tests/test_gpu_longprompt.py)"}


# ------------------------------------------------------------------------------------------------------------------
# stub runner (CPU): same control flow through the distributed code, scripted timings
# ------------------------------------------------------------------------------------------------------------------
class _StubModel:
    pass


def _stub_utterance(seed):
    time.sleep(0.01)
    rng = np.random.default_rng(seed)
    return 0.004 + 0.001 * (seed % 3), 0.01, FRAMES, rng.standard_normal(1000 + seed % 7).astype(np.float32)


def self_launch(n: int) -> int:
    """Re-run this command line as `n` ranks: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1
    --master-port <free> bench.py <same arguments>`.  Returns the launcher's exit code (non-zero when any rank fails, e.g. when
    the box has fewer than `n` GPUs)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only (profiling runs)")
    ap.add_argument("--batch", type=int, default=128, help="lock-step lanes of the batched figures, <= 128 (0 = skip them)")
    ap.add_argument("--batch-groups", type=int, default=1, help="opt-in: concurrent lock-step batches on one GPU, `batched_groups_one_gpu` (1 = skip; measured: two host-threaded groups of 8 give 159x vs 159x for one)")
    ap.add_argument("--batch-lookahead", type=int, default=None, help="measurement switch: 0 = the batch scheduler waits for every batch of frames right after queuing it (default: it keeps one batch queued ahead)")
    ap.add_argument("--config3-utterances", type=int, default=64, help="utterances of the sharded batched run (0 = skip)")
    ap.add_argument("--concurrent", type=int, default=4,
                    help="extra figure (N=1 only, after the timed region): utterances in flight on one GPU (0 = skip)")
    ap.add_argument("--no-1p7b", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc FETCH_SIZE pass (roofline.traffic)")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    global BATCH_LOOKAHEAD
    BATCH_LOOKAHEAD = args.batch_lookahead
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` (no launcher): start the N ranks ourselves -- one process per GPU under torchrun, loopback
        # rendezvous -- and hand their exit code back.  Under torchrun (the driver's N > 1 form) this branch is never taken.
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); "
                         f"use `python bench.py --gpus {args.gpus}` or torchrun --nproc-per-node {args.gpus}")
    stub = args.stub
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (there is no CPU path for the product)")
    # FQ3_BENCH_BACKEND=gloo + FQ3_BENCH_ONE_DEVICE=1: smoke-test the N>1 code path on a 1-GPU box
    backend = os.environ.get("FQ3_BENCH_BACKEND", "gloo" if stub else "nccl")
    if os.environ.get("FQ3_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    device = "cpu" if stub else f"cuda:{local_rank}"
    if not stub:
        torch.cuda.set_device(local_rank)
    coll_dev = device if backend == "nccl" else "cpu"
    # how many ranks the COLLECTIVE library itself sees (RCCL on device tensors under backend "nccl"; gloo in the CPU test): an
    # all-reduce of ones, so that the line proves its own n_gpus instead of echoing WORLD_SIZE
    coll_ranks = 1
    if world > 1:
        ones = torch.ones(1, device=coll_dev, dtype=torch.float32)
        dist.all_reduce(ones)
        coll_ranks = int(round(float(ones.item())))
        if coll_ranks != args.gpus:
            raise SystemExit(f"bench.py: the {backend} all-reduce counted {coll_ranks} rank(s), --gpus says {args.gpus}")

    def barrier():
        if not stub:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    cfg = model = prompt = req = None
    frame_ms = p_mid = None
    if stub:
        run_one = lambda seed: _stub_utterance(seed)
    else:
        cfg, model = build_model(device, codec_precision=HEADLINE_CODEC)         # every figure: the 1e-3-compliant vocoder
        _c, model_bf16 = build_model(device, share=model)                        # same weight replica, bf16 vocoder (reported beside the headline)
        req = build_request(cfg, device)
        run_one = lambda seed: one_utterance(model, req, seed)

    # (a high-priority decode stream was tried: no single-stream gain, and it quarters the throughput of the
    #  concurrent-utterance mode -- profiles/r01_concurrent_streams.txt -- so everything stays on default-priority streams)
    for i in range(args.warmup):
        run_one(1000 + i)
    if not stub:
        prompt = prepared_prompt(model, req)
        frame_ms, p_mid = measure_frame_graph(model, prompt)

    barrier()
    t0 = time.perf_counter()
    ttfas, rtfs, frames_total, pcm = [], [], 0, np.zeros(0, np.float32)
    for i in range(args.steps):
        ttfa, wall, n, pcm = run_one(2000 + rank * 100 + i)
        ttfas.append(ttfa); rtfs.append(n * FRAME_S / wall); frames_total += n
    barrier()
    elapsed = time.perf_counter() - t0

    extras = {}
    solo = world == 1 and not stub and not args.no_extras

    def guarded(key, fn):
        try:
            extras[key] = fn()
        except Exception as e:          # an extra never costs the headline line
            extras[key] = {"error": repr(e)}

    if solo and args.concurrent > 1:
        guarded("concurrent_utterances_one_gpu", lambda: concurrent_throughput(cfg, model, req, device, streams=args.concurrent))
    if solo:
        def _bf16_codec_headline():
            one_utterance(model_bf16, req, 1500)
            r = [one_utterance(model_bf16, req, 1501 + i) for i in range(min(max(args.steps, 1), 5))]
            return {"vocoder": "bf16 (what the reference's Torch path runs; 7.6e-3 PCM RMS from the fp32 oracle, parity_pcm)",
                    "value": round(float(np.mean([n * FRAME_S / w for _t, w, n, _p in r])), 3), "unit": "x real-time",
                    "ttfa_ms_p50": round(1000 * float(np.median([t for t, _w, _n, _p in r])), 2), "utterances": len(r)}
        guarded("headline_with_bf16_codec", _bf16_codec_headline)
        guarded("reference_audio_analysis", lambda: ref_analysis_block(cfg, device))
        guarded("parity_bf16_frames", lambda: parity_note(cfg, model))
        guarded("parity_pcm", lambda: parity_pcm(cfg, model_bf16, model, device))
        guarded("roofline_mfma", lambda: measure_mfma(cfg, model_bf16, prompt))
    if solo and args.batch > 1:
        def _batched():
            lanes = min(args.batch, 128)
            batched_run(model, prompt, lanes, lanes)                                   # warm-up: contexts, graph capture
            audio_s, wall, _ = batched_run(model, prompt, 2 * lanes, lanes)
            ms, p = batched_frame_time(model, cfg, prompt, lanes)
            out = {"lanes": lanes, "utterances": 2 * lanes, "value": round(audio_s / wall, 3),
                   "unit": "x real-time (aggregate audio s / wall s, one GPU, prefill + lock-step decode + non-streaming vocoder, utterances that finish together vocoded as one batch)",
                   "ms_per_lockstep_frame": round(ms, 3), "decode_only_value": round(lanes * 80.0 / ms, 1),
                   "end_to_end_over_decode_only": round((audio_s / wall) / (lanes * 80.0 / ms), 3),
                   "roofline": frame_roofline(cfg, ms, p, lanes=lanes, what=f"batched decode-frame hipGraph, {lanes} lanes, matrix-core GEMVs (default)")}
            try:
                st = model._batch_decoder(lanes).kv_pool.stats()
                out["kv_pool"] = {"blocks": st["blocks"], "high_water_blocks": st["high_water"], "mb_per_block": round(st["bytes_per_block"] / 2 ** 20, 2),
                                  "note": "paged talker KV: 64-key blocks from one pool; high_water = most blocks in use at once over the runs above"}
            except Exception as e:
                out["kv_pool"] = {"error": repr(e)}
            if not args.no_pmc:
                tr, how = measure_frame_traffic(lanes=lanes)
                if tr is not None:
                    out["roofline"]["traffic"] = float(round(tr))
                    out["roofline"]["traffic_over_algorithmic"] = round(tr / out["roofline"]["algorithmic_bytes_per_launch"], 3)
                    out["roofline"]["traffic_source"] = f"measured in this run: rocprofv3 --pmc FETCH_SIZE over tools/pmc_workload.py batch {lanes} (own pass, x2 gfx950 correction, {how})"
                else:
                    out["roofline"]["traffic"] = None
                    out["roofline"]["traffic_source"] = f"in-run pass failed: {how}"
            try:
                batched_streaming_run(model, req, lanes, lanes)                        # warm-up
                out["streaming"] = batched_streaming_run(model, req, lanes, 2 * lanes)
            except Exception as e:
                out["streaming"] = {"error": repr(e)}
            # first-chunk latency of SIMULTANEOUS requests at the smaller lane counts too (the bar of the single stream, < 150 ms, holds up
            # to ~50 simultaneous requests: ~35 ms + ~2 ms per request -- prompt build, packed prefill, first-chunk vocoder)
            for other in (32, 64):
                if lanes > other:
                    try:
                        batched_streaming_run(model, req, other, other)
                        out[f"streaming_{other}_lanes"] = batched_streaming_run(model, req, other, 2 * other)
                    except Exception as e:
                        out[f"streaming_{other}_lanes"] = {"error": repr(e)}
            try:
                ms2, p2 = batched_frame_time(model, cfg, prompt, lanes, mfma=0)
                out["valu_gemv"] = {"ms_per_lockstep_frame": round(ms2, 3), "decode_only_value": round(lanes * 80.0 / ms2, 1),
                                    "note": "VALU batch GEMVs: lanes bit-identical to single-stream decoding"}
            except Exception as e:
                out["valu_gemv"] = {"error": repr(e)}
            for other in (16, 32, 64, 128):
                if other == lanes:
                    continue
                try:
                    mso, _po = batched_frame_time(model, cfg, prompt, other)
                    out[f"lanes_{other}"] = {"ms_per_lockstep_frame": round(mso, 3), "decode_only_value": round(other * 80.0 / mso, 1)}
                except Exception as e:
                    out[f"lanes_{other}"] = {"error": repr(e)}
            return out
        guarded("batched_decode_one_gpu", _batched)
        if args.batch_groups > 1:
            guarded("batched_groups_one_gpu", lambda: batched_groups_run(cfg, model, prompt, device, args.config3_utterances or 64,
                                                                         groups=args.batch_groups, lanes=min(args.batch, 128)))

    # ---- BASELINE configs[3]: 1.7B-CustomVoice shapes, 64 utterances sharded over the ranks, 16 lock-step lanes per GPU (all
    #      ranks take part), through the public generate_custom_voice_batch ----
    c3 = None
    cfg17 = model17 = None
    if args.config3_utterances > 0 and args.batch > 1 and not args.no_extras:
        from fq3hip.sharding import shard_indices
        mine = shard_indices(args.config3_utterances, rank, world)
        # one wave: as many lanes as this rank has utterances (a lock-step frame costs what its token tiles cost, filled or not)
        lanes = min(args.batch, 128, max(16, 16 * ((len(mine) + 15) // 16)))
        err, c3_audio, c3_lens = None, 0.0, []
        texts = sentences(args.config3_utterances)
        try:
            if not stub:
                cfg17, model17 = build_model(device, "1p7b", max_seq_len=6144 if world == 1 else 2048, model_type="custom_voice",
                                             codec_precision=HEADLINE_CODEC)
                custom_voice_batch_run(model17, [texts[i] for i in mine[:lanes]], lanes, frames=16, seed=1999)     # warm-up: contexts, graph
        except Exception as e:
            err = repr(e)
        barrier()
        tc = time.perf_counter()
        if err is None:
            try:
                if stub:
                    time.sleep(0.005 * len(mine))
                    c3_audio, c3_lens = len(mine) * FRAMES * FRAME_S, [1000 + i for i in mine]
                else:
                    c3_audio, _w, c3_lens = custom_voice_batch_run(model17, [texts[i] for i in mine], lanes, seed=2000 + rank)
            except Exception as e:
                err = repr(e)
        barrier()
        c3 = {"audio_s": c3_audio, "wall": time.perf_counter() - tc, "lens": c3_lens, "lanes": lanes}
        if err is not None:
            c3["error"] = err

    # ---- reductions: max wall, summed frames, ONE gather of the per-rank results -----------------------------------------
    n_gathered = len(pcm)
    # every rank's own figures (its utterances over ITS wall time; its median first-chunk latency), so that at N = 1 the line's value
    # IS per_rank.value[0] and at N > 1 a slow rank shows
    per_rank = {"value": [round(frames_total * FRAME_S / elapsed, 3)], "ttfa_ms_p50": [round(1000 * float(np.median(ttfas)), 2) if ttfas else None]}
    if world > 1:
        mine = torch.tensor([frames_total * FRAME_S / elapsed, 1000 * float(np.median(ttfas)) if ttfas else -1.0], device=coll_dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = {"value": [round(float(t[0]), 3) for t in every], "ttfa_ms_p50": [round(float(t[1]), 2) for t in every]}
    if world > 1:
        from fq3hip.sharding import gather_arrays
        stats = torch.tensor([elapsed, c3["wall"] if c3 else 0.0], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        elapsed = float(stats[0])
        sums = torch.tensor([frames_total, c3["audio_s"] if c3 else 0.0], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(sums)
        frames_total = int(sums[0])
        # result gather of the north star, one collective pair: [n_ttfa, ttfas..., n_c3, c3 lens..., pcm...] per rank
        payload = np.concatenate([[len(ttfas)], np.asarray(ttfas, np.float32), [len(c3["lens"]) if c3 else 0],
                                  np.asarray(c3["lens"] if c3 else [], np.float32), pcm.astype(np.float32)]).astype(np.float32)
        parts = gather_arrays(payload, coll_dev)
        ttfas, c3_lens_all, n_gathered = [], [], 0
        for a in parts:
            k = int(a[0]); ttfas += [float(x) for x in a[1:1 + k]]
            j = int(a[1 + k]); c3_lens_all += [int(x) for x in a[2 + k:2 + k + j]]
            n_gathered += len(a) - (2 + k + j)
        if c3:
            c3 = dict(c3, wall=float(stats[1]), audio_s=float(sums[1]), lens=c3_lens_all)

    if rank == 0:
        audio_s = frames_total * FRAME_S
        value = audio_s / elapsed
        out = {
            "metric": "real-time factor (audio s / wall s), Qwen3-TTS-12Hz-0.6B voice-clone streaming chunk_size=8; p50 TTFA in ttfa_ms_p50",
            "value": round(value, 3), "unit": "x real-time", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / max(args.steps, 1), 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (seeded random weights at 0.6B shapes incl. text embedding / projection; 200-row ICL prompt built from text + 170 synthetic reference frames)",
            "config": {"workload": "configs[1]: Qwen3-TTS-12Hz-0.6B-Base voice-clone streaming chunk_size=8, hipGraph decode",
                       "vocoder": f"codec_precision={HEADLINE_CODEC!r}: the vocoder mode that meets the 1e-3 PCM bound (parity_pcm); the same run with the "
                                  "bf16 vocoder is in headline_with_bf16_codec",
                       "prompt_tokens": PROMPT_LEN, "ref_frames": REF_FRAMES, "frames_per_utterance": FRAMES,
                       "timed_region": "public generate_voice_clone_streaming(): tokenisation (byte-level stand-in tokenizer: no HF tokenizer "
                                       "offline) + prompt build (HIP) + prefill + decode + streaming vocoder, as the reference times it "
                                       "(README.md:219); weights, reference codes and speaker embedding are HBM-resident",
                       "utterances_per_gpu": args.steps, "sampling": "T=0.9 top_k=50 top_p=1.0 rep=1.05 (predictor T=0.9 top_k=50)",
                       "parallelism": f"utterance-sharded x{world} (replicas, result gather only)"},
            "ttfa_ms_p50": round(1000 * float(np.median(ttfas)), 2), "ttfa_ms_mean": round(1000 * float(np.mean(ttfas)), 2),
            "rtf_single_stream_mean": round(float(np.mean(rtfs)), 3), "gathered_samples": int(n_gathered),
            "rccl_ranks": coll_ranks, "collective_backend": backend if world > 1 else None, "per_rank": per_rank,
            "reference_published": {"rtx4090_rtf": 4.78, "rtx4090_ttfa_ms": 156, "h100_rtf": 3.884, "h100_ttfa_ms": 228,
                                    "source": "reference README.md:227,229 (CUDA graphs, other hardware)"},
        }
        if stub:
            out["data"] = "stub (CPU control-flow test of the N > 1 branch; no kernel ran)"
        if frame_ms is not None:
            out["decode_ms_per_frame"] = round(frame_ms, 4)
            rl = frame_roofline(cfg, frame_ms, p_mid, what="decode-frame hipGraph (554 launches: predictor M=2 prefill pass + 14 token "
                                                           "passes, talker 28 layers, heads, samplers)")
            # HBM-side traffic of one frame: a separate `rocprofv3 --pmc FETCH_SIZE` pass (x2 gfx950 correction) run from here after
            # the timed region (N = 1 only); the 15 predictor weight passes re-read their 157 MB through the fabric every pass
            traffic, how = (None, "not collected (N > 1, stub, or --no-pmc)")
            if solo and not args.no_pmc:
                traffic, how = measure_frame_traffic()
            if traffic is not None:
                rl["traffic"] = float(round(traffic))
                rl["traffic_over_algorithmic"] = round(traffic / rl["algorithmic_bytes_per_launch"], 3)
                rl["traffic_source"] = f"measured in this run: rocprofv3 --pmc FETCH_SIZE over tools/pmc_workload.py frames (own pass, x2 gfx950 correction, {how})"
            else:
                rl["traffic"] = 3.455e9
                rl["traffic_source"] = f"measured_offline: profiles/r02_pmc_fetch_size.txt (rocprofv3 --pmc FETCH_SIZE, own pass, x2 gfx950 correction, 24 profiled frames); in-run pass: {how}"
            out["roofline"] = rl
        if c3 is not None:
            if "error" in c3:
                out["config3_sharded_batched"] = {"error": c3["error"]}
            else:
                out["config3_sharded_batched"] = {
                    "workload": f"configs[3]: Qwen3-TTS-12Hz-1.7B-CustomVoice shapes, {args.config3_utterances} utterances x {FRAMES} frames through "
                                f"generate_custom_voice_batch (speaker-id prompts, text in -> waveforms on the host), round-robin over "
                                f"{world} GPU(s), {c3['lanes']} lock-step lanes per GPU, non-streaming vocoder, result gather only",
                    "value": round(c3["audio_s"] / c3["wall"], 3) if c3["wall"] > 0 else None,
                    "unit": "x real-time (aggregate audio s / max wall s)", "utterances": len(c3["lens"]), "wall_s": round(c3["wall"], 3)}
        out.update(extras)
        if solo and not args.no_1p7b:
            try:
                if model17 is None:
                    cfg17, model17 = build_model(device, "1p7b", max_seq_len=6144, model_type="custom_voice", codec_precision=HEADLINE_CODEC)
                inner17 = model17.model.model
                inner17.tts_model_type = "base"                # configs[2] is the Base model: same weights, voice-clone entry points
                out["model_1p7b"] = model_1p7b_block(cfg17, model17, device, lanes=min(max(args.batch, 8), 128))
            except Exception as e:
                out["model_1p7b"] = {"error": repr(e)}
        if world == 1 and not stub and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:          # the baseline is a reported extra; never lose the GPU line to it
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
