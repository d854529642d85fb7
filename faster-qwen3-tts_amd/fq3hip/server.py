#!/usr/bin/env python3
"""OpenAI-compatible speech endpoint over the MI355X path: ``POST /v1/audio/speech`` with the request schema, formats,
voice registry and streaming-WAV behaviour of the reference's ``examples/openai_server.py`` (:77-265).

What differs is the scheduling.  The reference serialises requests with one lock (``openai_server.py:71``) because its graph
objects hold a single static context.  Here there are two schedulers:

* ``lock``   one request at a time, audio streamed chunk by chunk as it is generated (the reference's behaviour);
* ``batch``  a worker thread owns the model and runs the continuous-batching decoder (``fq3hip/batching.py`` over
             ``fq3_batch_*``): requests that arrive while others are decoding join at the next frame boundary, up to
             ``lanes`` (<= 128) utterances advance in lock-step over ONE pass of the weights per frame; each response is
             sent when its utterance finishes.

``create_app(model, voices, ...)`` is the testable core; ``main()`` is the command line (same flags as the reference plus
``--scheduler/--lanes/--voice-cache/--synthetic``)."""
from __future__ import annotations

import argparse
import asyncio
import json
import logging
import os
import queue
import sys
import threading
from typing import Any, AsyncGenerator, Dict, Optional

import numpy as np
from pydantic import BaseModel

from .audio_io import to_pcm16, to_wav_bytes, wav_header

logger = logging.getLogger(__name__)
CONTENT_TYPES = {"wav": "audio/wav", "pcm": "audio/pcm", "mp3": "audio/mpeg"}


class SpeechRequest(BaseModel):
    """examples/openai_server.py:77-82."""
    model: str = "tts-1"
    input: str
    voice: str = "alloy"
    response_format: str = "wav"          # wav | pcm | mp3
    speed: float = 1.0                     # accepted, not applied (as in the reference)


class BatchWorker:
    """One thread owns the model and runs the continuous-batching decoder in streaming mode.  ``submit`` returns a queue that
    receives the utterance's PCM chunks (``np.ndarray``) as they are vocoded, then ``BatchWorker.DONE`` -- or an exception."""

    DONE = object()

    def __init__(self, model, lanes: int = 8, chunk_size: int = 12):
        from .batching import MAX_LANES
        self.model, self.lanes = model, max(1, min(int(lanes), MAX_LANES))
        # ONE chunk size for the whole server: the lock-step decoder hands out chunks of this many frames and every request's
        # StreamingVocoder is built for the same number (a voice entry's own "chunk_size" only applies to the lock scheduler)
        self.chunk_size = max(1, int(chunk_size))
        self.inbox: "queue.Queue" = queue.Queue()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def submit(self, voice_cfg: dict, text: str) -> "queue.Queue":
        out: "queue.Queue" = queue.Queue()
        self.inbox.put((voice_cfg, text, out))
        return out

    def _run(self):
        import torch
        from .batching import BatchRequest
        m = self.model
        while True:
            first = self.inbox.get()
            if first is None:
                return
            waiting: Dict[int, Any] = {}
            counter = [0]
            try:
                with torch.inference_mode():
                    def prepare(item):
                        """(voice cfg, text, reply queue) -> BatchRequest, or None when the request failed before decoding."""
                        cfg, text, out = item
                        try:
                            inner, talker, config, tie, tam, tth, tpe, rc = m._prepare_generation(
                                text=text, language=cfg.get("language", "Auto"), ref_audio=cfg.get("ref_audio"),
                                ref_text=cfg.get("ref_text", ""), voice_clone_prompt=cfg.get("voice_clone_prompt"),
                                non_streaming_mode=False)
                        except Exception as exc:
                            out.put(exc)
                            return None
                        i = counter[0]
                        counter[0] += 1
                        waiting[i] = (out, m.streaming_vocoder(rc, self.chunk_size))
                        kw = m._gen_kwargs(int(cfg.get("max_new_tokens", 2048)), 2, 0.9, 50, 1.0, True, 1.05)
                        return BatchRequest(i, talker, tie, tam, tth, tpe, config, kw)

                    def source():
                        """Polled by the scheduler at every frame boundary: requests that arrived while the batch was decoding."""
                        while True:
                            try:
                                item = self.inbox.get_nowait()
                            except queue.Empty:
                                return None
                            if item is None:                   # shutdown marker: finish what is running, then stop
                                self.inbox.put(None)
                                return None
                            req = prepare(item)
                            if req is not None:
                                return req

                    head = prepare(first)
                    chunk_frames = self.chunk_size
                    for rid, codes, info in m._batch_decoder(self.lanes).run([head] if head is not None else [], on_error="yield",
                                                                             source=source, chunk_frames=chunk_frames):
                        out, voc = waiting[rid]
                        final = bool(info.get("is_final")) or "error" in info
                        if "error" in info or (codes is None and final and info.get("steps", 0) == 0):
                            out.put(RuntimeError(info.get("error", "generation returned no tokens")))
                        elif codes is not None and codes.shape[0] > 0:
                            audio, _sr = voc.push(codes, info.get("codes_ready_event"))
                            out.put(np.asarray(audio, dtype=np.float32))
                        if final:
                            out.put(self.DONE)
                            waiting.pop(rid, None)
                # the scheduler is done: nobody may be left without an answer (a request it never reported would otherwise
                # block its HTTP handler and an executor thread forever)
                for out, _ in waiting.values():
                    out.put(RuntimeError("the batch scheduler finished without an answer for this request"))
                    out.put(self.DONE)
                waiting.clear()
            except Exception as exc:            # a failed batch answers everyone who is still waiting
                for out, _ in waiting.values():
                    out.put(exc)
                    out.put(self.DONE)
                waiting.clear()


def create_app(model, voices: Dict[str, dict], default_voice: Optional[str] = None, scheduler: str = "lock", lanes: int = 8,
               chunk_size: int = 12):
    from fastapi import FastAPI, HTTPException
    from fastapi.responses import Response, StreamingResponse

    app = FastAPI(title="faster-qwen3-tts (MI355X) OpenAI-compatible API")
    lock = threading.Lock()
    worker = BatchWorker(model, lanes, chunk_size) if scheduler == "batch" else None
    sample_rate = int(getattr(model, "sample_rate", 24000))

    def resolve_voice(name: str) -> dict:
        if name in voices:
            return voices[name]
        if default_voice and default_voice in voices:
            logger.warning("Voice %r not configured; falling back to default voice %r", name, default_voice)
            return voices[default_voice]
        raise HTTPException(status_code=400, detail=f"Voice {name!r} is not configured. Available voices: {list(voices.keys())}")

    def clone_kwargs(cfg: dict, text: str) -> dict:
        return dict(text=text, language=cfg.get("language", "Auto"), ref_audio=cfg.get("ref_audio"), ref_text=cfg.get("ref_text", ""),
                    voice_clone_prompt=cfg.get("voice_clone_prompt"))

    async def stream_chunks(cfg: dict, text: str) -> AsyncGenerator[bytes, None]:
        q: "queue.Queue" = queue.Queue()
        done = object()

        def producer():
            try:
                with lock:
                    for chunk, _sr, _t in model.generate_voice_clone_streaming(chunk_size=cfg.get("chunk_size", 12),
                                                                               non_streaming_mode=False, **clone_kwargs(cfg, text)):
                        q.put(chunk)
            except Exception as exc:
                q.put(exc)
            finally:
                q.put(done)

        threading.Thread(target=producer, daemon=True).start()
        loop = asyncio.get_event_loop()
        while True:
            item = await loop.run_in_executor(None, q.get)
            if item is done:
                break
            if isinstance(item, Exception):
                raise item
            yield to_pcm16(item)

    @app.get("/health")
    async def health():
        return {"status": "ok", "model_loaded": model is not None, "scheduler": scheduler, "lanes": lanes if worker else 1}

    @app.post("/v1/audio/speech")
    async def create_speech(req: SpeechRequest):
        if model is None:
            raise HTTPException(status_code=503, detail="Model not loaded")
        if not req.input.strip():
            raise HTTPException(status_code=400, detail="'input' text is empty")
        cfg = resolve_voice(req.voice)
        fmt = req.response_format.lower()
        if fmt not in CONTENT_TYPES:
            raise HTTPException(status_code=400, detail=f"response_format {fmt!r} not supported. Use: wav, pcm, mp3")
        if fmt == "mp3":
            raise HTTPException(status_code=400, detail="response_format='mp3' needs pydub + ffmpeg, which this image does not ship; use wav or pcm")
        loop = asyncio.get_event_loop()
        if worker is not None:
            box = worker.submit(cfg, req.input)

            async def batch_stream():
                first = True
                while True:
                    item = await loop.run_in_executor(None, box.get)
                    if item is BatchWorker.DONE:
                        break
                    if isinstance(item, Exception):
                        if first:
                            raise HTTPException(status_code=500, detail=repr(item))
                        logger.error("generation failed mid-stream: %r", item)
                        break
                    if first and fmt == "wav":
                        yield wav_header(sample_rate)      # unknown data length: streaming
                    first = False
                    yield to_pcm16(item)

            # pull the first event before answering, so that a request that fails outright is a 500, not an empty 200
            gen = batch_stream()
            try:
                head = await gen.__anext__()
            except StopAsyncIteration:
                head = None

            async def replay():
                if head is not None:
                    yield head
                async for raw in gen:
                    yield raw

            return StreamingResponse(replay(), media_type=CONTENT_TYPES[fmt])

        async def audio_stream():
            if fmt == "wav":
                yield wav_header(sample_rate)          # unknown data length: streaming
            async for raw in stream_chunks(cfg, req.input):
                yield raw

        return StreamingResponse(audio_stream(), media_type=CONTENT_TYPES[fmt])

    return app


def main(argv=None):
    p = argparse.ArgumentParser(description="OpenAI-compatible TTS server over the MI355X HIP path")
    p.add_argument("--model", default=os.environ.get("QWEN_TTS_MODEL", "Qwen/Qwen3-TTS-12Hz-1.7B-Base"))
    p.add_argument("--voices", default=os.environ.get("QWEN_TTS_VOICES"), metavar="FILE")
    p.add_argument("--ref-audio", default=os.environ.get("QWEN_TTS_REF_AUDIO"), metavar="FILE")
    p.add_argument("--ref-text", default=os.environ.get("QWEN_TTS_REF_TEXT", ""))
    p.add_argument("--language", default=os.environ.get("QWEN_TTS_LANGUAGE", "Auto"))
    p.add_argument("--host", default="0.0.0.0")
    p.add_argument("--port", type=int, default=8000)
    p.add_argument("--device", default="cuda")
    p.add_argument("--scheduler", default="batch", choices=["lock", "batch"])
    p.add_argument("--lanes", type=int, default=8)
    p.add_argument("--chunk-size", type=int, default=12, help="frames per streamed chunk (batch scheduler: server-wide)")
    p.add_argument("--voice-cache", help="directory of precomputed voice prompts serving ref_audio entries")
    p.add_argument("--synthetic", choices=["0.6b", "1.7b"])
    args = p.parse_args(argv)
    if args.voices:
        with open(args.voices) as f:
            voices = json.load(f)
        default_voice = next(iter(voices))
    elif args.ref_audio:
        voices = {"default": {"ref_audio": args.ref_audio, "ref_text": args.ref_text, "language": args.language}}
        default_voice = "default"
    else:
        print("ERROR: provide --ref-audio <file> or --voices <config.json>", file=sys.stderr)
        sys.exit(1)
    from types import SimpleNamespace
    from .cli import load_model
    model = load_model(SimpleNamespace(model=args.model, device=args.device, dtype="bf16", backend="torch", synthetic=args.synthetic,
                                       voice_cache=args.voice_cache))
    import uvicorn
    logging.basicConfig(level=logging.INFO)
    uvicorn.run(create_app(model, voices, default_voice, args.scheduler, args.lanes, args.chunk_size), host=args.host, port=args.port)


if __name__ == "__main__":
    main()
