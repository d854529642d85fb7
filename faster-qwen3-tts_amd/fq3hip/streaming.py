"""Streaming decode with the reference's generator contract (``faster_qwen3_tts/streaming.py:19-188``):
yields ``(codec_chunk LongTensor[chunk_steps, 16], timing)`` every ``chunk_size`` frames; timing keys
``chunk_index, chunk_steps, prefill_ms (first chunk only), decode_ms, total_steps_so_far, is_final``.
One host sync per chunk (the reference also syncs per chunk, plus ``.item()`` per frame)."""
from __future__ import annotations

import time
from typing import Generator, Tuple

import torch

from .generate import _prefill_and_arm, run_frames
from .predictor_graph import PredictorGraph
from .talker_graph import TalkerGraph


@torch.inference_mode()
def fast_generate_streaming(talker, talker_input_embeds: torch.Tensor, attention_mask: torch.Tensor,
                            trailing_text_hiddens: torch.Tensor, tts_pad_embed: torch.Tensor, config,
                            predictor_graph: PredictorGraph, talker_graph: TalkerGraph, max_new_tokens: int = 2048,
                            min_new_tokens: int = 2, temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0,
                            do_sample: bool = True, repetition_penalty: float = 1.05, chunk_size: int = 12,
                            use_graph: bool = True) -> Generator[Tuple[torch.Tensor, dict], None, None]:
    t_start = time.time()
    eng, tn, pn, max_frames = _prefill_and_arm(
        talker, talker_input_embeds, attention_mask, trailing_text_hiddens, tts_pad_embed, config, predictor_graph,
        talker_graph, max_new_tokens, min_new_tokens, temperature, top_k, top_p, do_sample, repetition_penalty,
        use_graph=use_graph)
    torch.cuda.current_stream(eng.device).synchronize()
    t_prefill = time.time() - t_start
    issued, emitted, chunk_count, done = 0, 0, 0, False
    chunk_start = time.time()
    issued = run_frames(eng, tn, pn, issued, min(chunk_size, max_frames))
    while True:
        n, done = eng.decode_poll()              # host sync: chunk k is complete
        new = n - emitted
        if new <= 0:
            break
        chunk = eng.decode_codes(emitted, new)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(eng.device))
        emitted = n
        is_final = new < chunk_size              # streaming.py:157-188: only a trailing partial chunk is final
        more = (not done) and (not is_final) and issued < max_frames
        if more:
            # look-ahead: chunk k+1 decodes while the consumer vocodes chunk k on its own stream
            issued = run_frames(eng, tn, pn, issued, min(chunk_size, max_frames - issued))
        yield chunk, {
            "chunk_index": chunk_count,
            "chunk_steps": new,
            "prefill_ms": t_prefill * 1000 if chunk_count == 0 else 0,
            "decode_ms": (time.time() - chunk_start) * 1000,
            "total_steps_so_far": emitted,
            "is_final": is_final,
            "codes_ready_event": ready,          # extra key: lets a consumer on another stream wait for `chunk` only
        }
        chunk_count += 1
        chunk_start = time.time()
        if not more:
            break


def parity_generate_streaming(talker, talker_input_embeds, attention_mask, trailing_text_hiddens, tts_pad_embed, config,
                              max_new_tokens: int = 2048, min_new_tokens: int = 2, temperature: float = 0.9,
                              top_k: int = 50, top_p: float = 1.0, do_sample: bool = True,
                              repetition_penalty: float = 1.05, chunk_size: int = 12, predictor_graph=None,
                              talker_graph=None):
    """The reference's parity streaming (streaming.py:192-359) replays the upstream dynamic-cache model
    without graphs.  The equivalent here is the same HIP kernels as direct launches (no hipGraph)."""
    if predictor_graph is None or talker_graph is None:
        raise ValueError("parity_generate_streaming needs the graph objects of the HIP context")
    return fast_generate_streaming(talker, talker_input_embeds, attention_mask, trailing_text_hiddens, tts_pad_embed,
                                   config, predictor_graph, talker_graph, max_new_tokens=max_new_tokens,
                                   min_new_tokens=min_new_tokens, temperature=temperature, top_k=top_k, top_p=top_p,
                                   do_sample=do_sample, repetition_penalty=repetition_penalty, chunk_size=chunk_size,
                                   use_graph=False)
