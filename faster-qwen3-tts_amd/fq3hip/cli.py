#!/usr/bin/env python3
"""Command line for the MI355X path, with the reference CLI's sub-commands and flags (``faster_qwen3_tts/cli.py``):
``clone`` / ``custom`` / ``design`` synthesise one text to a WAV file, ``serve`` reads one text per stdin line
(``cli.py:228-349``).  Differences: WAV files are written with the standard library (no ``soundfile`` in this image);
``--voice-cache DIR`` serves ``--ref-audio`` from precomputed voice prompts (``fq3hip/voice_cache.py``), because
reference-audio analysis is not part of this path; ``--synthetic 0.6b|1.7b`` builds seeded random weights instead of
loading a checkpoint (smoke runs, throughput measurements); ``serve --lanes N`` decodes up to N queued lines in lock-step
(``fq3_batch_*``); GGML flags are accepted and refused like the wrapper refuses them."""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np


def _stream_to_audio(gen):
    chunks, sr = [], None
    for audio_chunk, sr, _ in gen:
        chunks.append(audio_chunk)
    if not chunks:
        return np.zeros(1, dtype=np.float32), 24000
    return np.concatenate(chunks), sr


def load_model(args):
    """``cli.py:14-45``."""
    import torch
    from .model import FasterQwen3TTS
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}.get(args.dtype)
    if dtype is None:
        raise SystemExit("ERROR: --dtype fp16 is not built on this path (bf16 or fp32)")
    if args.backend != "torch":
        return FasterQwen3TTS.from_pretrained(args.model, backend=args.backend)       # raises NotImplementedError, as documented
    if args.synthetic:
        from .config import qwen3_tts_0p6b, qwen3_tts_1p7b
        from .weights import synth_weights
        cfg = qwen3_tts_0p6b() if args.synthetic == "0.6b" else qwen3_tts_1p7b()
        cfg.spk_id = {"synthetic": cfg.talker.vocab_size - 1024 + 300}
        cfg.spk_is_dialect = {"synthetic": False}
        W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor", "codec", "text"), codec_normalized=True)
        model = FasterQwen3TTS.from_weights(cfg, W, device=args.device, dtype=dtype, max_seq_len=2048)
    else:
        model = FasterQwen3TTS.from_pretrained(args.model, device=args.device, dtype=dtype, attn_implementation="sdpa",
                                               max_seq_len=2048)
    if getattr(args, "voice_cache", None):
        model.set_voice_ref_cache(args.voice_cache)
    return model


def validate_clone_refs(args):
    """``cli.py:63-78`` (the cached-reference flags belong to the GGML backend there; here --voice-cache plays that role)."""
    if args.ref_spk or args.ref_rvq:
        print("ERROR: --ref-spk/--ref-rvq belong to the GGML backend; use --voice-cache DIR with --ref-audio")
        sys.exit(2)
    if not args.ref_audio:
        print("ERROR: clone mode requires --ref-audio")
        sys.exit(2)
    if not args.xvec_only and not args.ref_text and not args.voice_cache:
        print("ERROR: --ref-text is required with --ref-audio unless --xvec-only is set")
        sys.exit(2)


def _gen_kwargs(args):
    return dict(max_new_tokens=args.max_new_tokens, temperature=args.temperature, top_k=args.top_k, do_sample=not args.greedy,
                repetition_penalty=args.repetition_penalty)


def synthesize(model, args, text: str):
    """One text -> (waveform, sample_rate) in the selected mode (``cli.py:81-225``)."""
    kw = _gen_kwargs(args)
    if args.mode == "clone":
        ckw = dict(text=text, language=args.language, ref_audio=args.ref_audio, ref_text=args.ref_text, xvec_only=args.xvec_only,
                   non_streaming_mode=args.non_streaming_mode, **kw)
        if args.streaming:
            return _stream_to_audio(model.generate_voice_clone_streaming(chunk_size=args.chunk_size, **ckw))
        audio_list, sr = model.generate_voice_clone(**ckw)
        return audio_list[0], sr
    if args.mode == "custom":
        ckw = dict(text=text, speaker=args.speaker, language=args.language, instruct=args.instruct, **kw)
        if args.streaming:
            return _stream_to_audio(model.generate_custom_voice_streaming(chunk_size=args.chunk_size, **ckw))
        audio_list, sr = model.generate_custom_voice(**ckw)
        return audio_list[0], sr
    ckw = dict(text=text, instruct=args.instruct, language=args.language, **kw)
    if args.streaming:
        return _stream_to_audio(model.generate_voice_design_streaming(chunk_size=args.chunk_size, **ckw))
    audio_list, sr = model.generate_voice_design(**ckw)
    return audio_list[0], sr


def _check_mode_args(args):
    if args.mode == "clone":
        validate_clone_refs(args)
    if args.mode == "custom" and not args.speaker:
        print("ERROR: --speaker is required for custom mode")
        sys.exit(2)
    if args.mode == "design" and not args.instruct:
        print("ERROR: --instruct is required for design mode")
        sys.exit(2)


def cmd_once(args, model=None):
    from .audio_io import write_wav
    _check_mode_args(args)
    model = model or load_model(args)
    start = time.perf_counter()
    audio, sr = synthesize(model, args, args.text)
    total = time.perf_counter() - start
    write_wav(args.output, audio, sr)
    dur = len(audio) / sr if sr else 0.0
    print(f"Wrote {args.output} (dur {dur:.2f}s, RTF {dur / total if total > 0 else 0.0:.2f})")


def cmd_serve(args, model=None, lines=None):
    """``cli.py:228-349``: one text per input line -> ``out_NNNN.wav``.  With ``--lanes N > 1`` (clone mode) the lines that are
    already waiting are decoded together, up to N in lock-step over one weight stream."""
    import os
    from .audio_io import write_wav
    _check_mode_args(args)
    model = model or load_model(args)
    print("Server started. Enter text per line. Type 'exit' or 'quit' to stop.")
    idx = 1
    it = iter(lines if lines is not None else sys.stdin)
    pending, stop = [], False
    while not stop:
        pending.clear()
        for line in it:
            text = line.strip()
            if not text:
                continue
            if text.lower() in ("exit", "quit", "stop"):
                stop = True
                break
            pending.append(text)
            if len(pending) >= max(1, args.lanes) or args.lanes <= 1:
                break
        else:
            stop = True
        if not pending:
            continue
        start = time.perf_counter()
        if len(pending) > 1 and args.mode == "clone":
            results = model.generate_voice_clone_batch(pending, language=args.language, ref_audio=args.ref_audio, ref_text=args.ref_text,
                                                       xvec_only=args.xvec_only, non_streaming_mode=args.non_streaming_mode,
                                                       lanes=args.lanes, **_gen_kwargs(args))
            outs = [(r[0][0], r[1]) for r in results]
        else:
            outs = [synthesize(model, args, t) for t in pending]
        total = time.perf_counter() - start
        for audio, sr in outs:
            out_path = os.path.join(args.output_dir, f"out_{idx:04d}.wav")
            idx += 1
            write_wav(out_path, audio, sr)
            dur = len(audio) / sr if sr else 0.0
            print(f"Wrote {out_path} (dur {dur:.2f}s, RTF {dur * len(outs) / total if total > 0 else 0.0:.2f})")


def build_parser():
    p = argparse.ArgumentParser(prog="faster-qwen3-tts", description="FasterQwen3TTS CLI (MI355X HIP path)")
    p.add_argument("--device", default="cuda", help="Device (a ROCm GPU: 'cuda' in PyTorch-ROCm)")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"], help="Model dtype")
    p.add_argument("--backend", default="torch", choices=["torch", "ggml"], help="Inference backend ('ggml' is refused on this path)")
    p.add_argument("--synthetic", choices=["0.6b", "1.7b"], help="seeded random weights at the real shapes instead of a checkpoint")
    p.add_argument("--voice-cache", help="directory of precomputed voice prompts (<key>.spk/.rvq/.json) serving --ref-audio")
    sub = p.add_subparsers(dest="command", required=True)

    def common(sp, output=True):
        sp.add_argument("--model", default="Qwen/Qwen3-TTS-12Hz-0.6B-Base", help="local checkpoint directory")
        sp.add_argument("--language", default="English")
        sp.add_argument("--max-new-tokens", type=int, default=2048)
        sp.add_argument("--temperature", type=float, default=0.9)
        sp.add_argument("--top-k", type=int, default=50)
        sp.add_argument("--repetition-penalty", type=float, default=1.05)
        sp.add_argument("--greedy", action="store_true")
        sp.add_argument("--streaming", action="store_true")
        sp.add_argument("--chunk-size", type=int, default=12)
        if output:
            sp.add_argument("--text", required=True)
            sp.add_argument("--output", required=True)

    def clone_refs(sp):
        sp.add_argument("--ref-audio")
        sp.add_argument("--ref-text", default="")
        sp.add_argument("--ref-spk")
        sp.add_argument("--ref-rvq")
        sp.add_argument("--xvec-only", action="store_true")
        sp.add_argument("--non-streaming-mode", action="store_true", default=None)

    c = sub.add_parser("clone", help="voice cloning from reference audio")
    common(c); clone_refs(c); c.set_defaults(mode="clone", func=cmd_once)
    c = sub.add_parser("custom", help="CustomVoice model: predefined speaker")
    common(c); c.add_argument("--speaker"); c.add_argument("--instruct"); c.set_defaults(mode="custom", func=cmd_once)
    c = sub.add_parser("design", help="VoiceDesign model: voice from an instruction")
    common(c); c.add_argument("--instruct"); c.set_defaults(mode="design", func=cmd_once)
    s = sub.add_parser("serve", help="read one text per stdin line, write out_NNNN.wav")
    common(s, output=False); clone_refs(s)
    s.add_argument("--mode", default="clone", choices=["clone", "custom", "design"])
    s.add_argument("--speaker"); s.add_argument("--instruct")
    s.add_argument("--output-dir", default="outputs")
    s.add_argument("--lanes", type=int, default=1, help="decode up to N waiting lines in lock-step (clone mode)")
    s.set_defaults(func=cmd_serve)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    args.func(args)


if __name__ == "__main__":
    main()
