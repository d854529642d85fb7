"""Non-streaming decode loop with the reference's signature and return contract
(``faster_qwen3_tts/generate.py:16-215``), driven by the fused on-device loop of ``libfq3hip``.

Per frame the reference does: ``token.item()`` host sync, ~60 eager launches of glue, two graph
replays.  Here a frame is ONE hipGraph replay (predictor passes, 16-way embedding sum, talker step,
codec head, penalty + sampler, all state on the device) and the host polls a done flag every
``poll_every`` frames.
"""
from __future__ import annotations

import time
from typing import Optional, Tuple

import torch

from .predictor_graph import PredictorGraph
from .talker_graph import TalkerGraph

NOISE_RING = 64      # frames of pre-drawn Exp(1) variates per refill


def _refill(engine, talker_noise, pred_noise):
    if talker_noise is not None:
        talker_noise.exponential_(1)
    if pred_noise is not None:
        pred_noise.exponential_(1)


def _n_pad_of(attention_mask) -> int:
    """Left padding of a prompt (model.py:774-787: the mask's zeros).  The HIP prompt builder notes the count on the mask it creates
    (``fq3_n_pad``: a single prompt has none); a foreign mask is counted on the device, which is a host wait."""
    if attention_mask is None:
        return 0
    known = noted_n_pad(attention_mask)
    return known if known is not None else int((attention_mask[0] == 0).sum())


def noted_n_pad(attention_mask):
    """The padding count the prompt builder noted on a mask it created -- valid only while the mask has not been written to since
    (the note carries the tensor's version counter) -- or None."""
    note = getattr(attention_mask, "fq3_n_pad", None)
    if not (isinstance(note, tuple) and len(note) == 2):
        return None
    if note[1] is None:                               # noted on an inference-mode tensor (no version counter)
        return int(note[0])
    if attention_mask.is_inference() or note[1] != attention_mask._version:
        return None
    return int(note[0])


def _prefill_first_token(eng, talker_input_embeds, attention_mask, config, min_new_tokens, temperature, top_k, top_p, do_sample):
    """=== PREFILL (generate.py:107-134) === on ``eng``: KV rows written, first codebook-0 token sampled.
    Returns (token, past_hidden, prompt_rows, n_pad)."""
    dt, dev = eng.dtype, eng.device
    eos_id = config.codec_eos_token_id
    V = config.vocab_size
    n_pad = _n_pad_of(attention_mask)
    x = talker_input_embeds[0].to(device=dev, dtype=dt).contiguous()
    logits, hidden = eng.prefill(x, n_pad=n_pad)
    first_noise = torch.empty(V, dtype=dt, device=dev).exponential_(1) if do_sample else None
    token = eng.sample(logits, temperature=temperature, top_k=top_k, top_p=top_p, do_sample=do_sample,
                       sup_lo=max(0, V - 1024), sup_hi=V, keep_id=eos_id, suppress_eos=min_new_tokens > 0,
                       noise=first_noise)
    return int(token), hidden, int(x.shape[0]), n_pad


def _prefill_first_tokens_packed(engines, items):
    """``_prefill_first_token`` for several requests at once: ONE packed prefill (``fq3_prefill_batch``: one pass over the
    weights), then the first tokens sampled per request.  ``items[i]`` = the argument tuple ``_prefill_first_token`` takes
    after the engine.  Returns one ``(token, past_hidden, prompt_rows, n_pad)`` per request."""
    from .engine import Fq3Engine
    xs, pads = [], []
    for eng, (tie, tam, _config, *_rest) in zip(engines, items):
        pads.append(_n_pad_of(tam))
        xs.append(tie[0].to(device=eng.device, dtype=eng.dtype).contiguous())
    outs = Fq3Engine.prefill_batch(engines, xs, pads)
    toks = []
    for eng, x, (logits, hidden), (tie, tam, config, min_new_tokens, temperature, top_k, top_p, do_sample) in zip(engines, xs, outs, items):
        V = config.vocab_size
        first_noise = torch.empty(V, dtype=eng.dtype, device=eng.device).exponential_(1) if do_sample else None
        toks.append(eng.sample(logits, temperature=temperature, top_k=top_k, top_p=top_p, do_sample=do_sample,
                               sup_lo=max(0, V - 1024), sup_hi=V, keep_id=config.codec_eos_token_id,
                               suppress_eos=min_new_tokens > 0, noise=first_noise))
    return [(int(t), o[1], int(x.shape[0]), p) for t, o, x, p in zip(toks, outs, xs, pads)]      # the first int() waits for all


def _prefill_first_tokens_launch(engines, items, pads):
    """The LAUNCH half of :func:`_prefill_first_tokens_packed` (the batch scheduler's pipelined first wave): the packed prefill and the
    first-token samplers are queued on the current stream and nothing is waited for.  ``pads[i]``: the prompt's left padding (the
    caller counted all of them with one device round trip).  Returns ``[(token TENSOR on the device, past_hidden, prompt_rows, n_pad)]``;
    the caller turns the tensors into ints when it needs them (one copy for all)."""
    from .engine import Fq3Engine
    xs = [tie[0].to(device=eng.device, dtype=eng.dtype).contiguous() for eng, (tie, *_r) in zip(engines, items)]
    outs = Fq3Engine.prefill_batch(engines, xs, pads)
    res = []
    for eng, x, (logits, hidden), p, (tie, tam, config, min_new_tokens, temperature, top_k, top_p, do_sample) in zip(engines, xs, outs, pads, items):
        V = config.vocab_size
        first_noise = torch.empty(V, dtype=eng.dtype, device=eng.device).exponential_(1) if do_sample else None
        tok = eng.sample(logits, temperature=temperature, top_k=top_k, top_p=top_p, do_sample=do_sample,
                         sup_lo=max(0, V - 1024), sup_hi=V, keep_id=config.codec_eos_token_id,
                         suppress_eos=min_new_tokens > 0, noise=first_noise)
        res.append((tok, hidden, int(x.shape[0]), int(p)))
    return res


def _arm_decode(talker, config, token, hidden, n_rows, attention_mask, trailing_text_hiddens, tts_pad_embed, predictor_graph, talker_graph,
                max_new_tokens, min_new_tokens, temperature, top_k, top_p, do_sample, repetition_penalty, use_graph, n_pad=None):
    """Arms the on-device loop of ``talker_graph.engine`` behind a prefill whose KV rows are already in its cache.
    ``n_pad``: the prompt's left padding when the caller already counted it (the prefill did): counting it here from the device mask
    is a host wait for the CURRENT stream -- in a batch scheduler that is every lock-step frame queued ahead."""
    eng = talker_graph.engine
    dt, dev = eng.dtype, eng.device
    V = config.vocab_size
    prefill_len = talker_graph.prefill_kv(n_rows)
    rope_deltas = getattr(talker, "rope_deltas", None)
    if n_pad is not None:
        talker_graph.set_generation_state(attention_mask, rope_deltas, n_pad=int(n_pad))      # no device reduction over the mask
    else:
        talker_graph.set_generation_state(attention_mask, rope_deltas)
    need_pred_noise = bool(predictor_graph.do_sample)
    tn = torch.empty(NOISE_RING, V, dtype=dt, device=dev) if do_sample else None
    pn = (torch.empty(NOISE_RING, eng.cfg.num_code_groups - 1, eng.cfg.predictor.vocab_size, dtype=dt, device=dev)
          if need_pred_noise else None)
    tth = trailing_text_hiddens[0].to(device=dev, dtype=dt).contiguous()
    tpe = tts_pad_embed.reshape(-1).to(device=dev, dtype=dt).contiguous()
    gen_step = 0        # out.generation_step of the prefill (generate.py:122)
    max_frames = min(int(max_new_tokens), eng.max_frames)
    eng.decode_begin(first_token=int(token), prefill_len=prefill_len, gen_step=gen_step, past_hidden=hidden,
                     trailing_text=tth, tts_pad_embed=tpe, temperature=temperature, top_k=top_k, top_p=top_p,
                     do_sample=do_sample, repetition_penalty=repetition_penalty, min_new_tokens=min_new_tokens,
                     max_new_tokens=max_frames, talker_noise=tn, pred_noise=pn, noise_frames=NOISE_RING)
    if use_graph:
        eng.graph_capture()
    else:
        eng.graph_reset()
    return eng, tn, pn, max_frames


def _prefill_and_arm(talker, talker_input_embeds, attention_mask, trailing_text_hiddens, tts_pad_embed, config,
                     predictor_graph, talker_graph, max_new_tokens, min_new_tokens, temperature, top_k, top_p,
                     do_sample, repetition_penalty, use_graph):
    token, hidden, n_rows, n_pad = _prefill_first_token(talker_graph.engine, talker_input_embeds, attention_mask, config,
                                                        min_new_tokens, temperature, top_k, top_p, do_sample)
    return _arm_decode(talker, config, token, hidden, n_rows, attention_mask, trailing_text_hiddens, tts_pad_embed, predictor_graph,
                       talker_graph, max_new_tokens, min_new_tokens, temperature, top_k, top_p, do_sample, repetition_penalty,
                       use_graph, n_pad=n_pad)


def run_frames(eng, tn, pn, issued: int, count: int):
    """Enqueue ``count`` more frames, refilling the noise rings on ring boundaries."""
    while count > 0:
        if issued % NOISE_RING == 0:
            _refill(eng, tn, pn)
        k = min(count, NOISE_RING - issued % NOISE_RING)
        eng.decode_frames(k)
        issued += k
        count -= k
    return issued


@torch.inference_mode()
def fast_generate(talker, talker_input_embeds: torch.Tensor, attention_mask: torch.Tensor,
                  trailing_text_hiddens: torch.Tensor, tts_pad_embed: torch.Tensor, config,
                  predictor_graph: PredictorGraph, talker_graph: TalkerGraph, max_new_tokens: int = 2048,
                  min_new_tokens: int = 2, temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0,
                  do_sample: bool = True, repetition_penalty: float = 1.05,
                  subtalker_dosample: Optional[bool] = None, subtalker_top_k: Optional[int] = None,
                  subtalker_top_p: Optional[float] = None, subtalker_temperature: Optional[float] = None,
                  parity_mode: bool = False, poll_every: int = 8) -> Tuple[Optional[torch.Tensor], dict]:
    """Returns (codec_ids LongTensor[T, 16] or None, timing dict with the reference's keys
    ``prefill_ms, decode_s, steps, ms_per_step, steps_per_s``).  ``parity_mode=True`` runs the same
    kernels as direct launches without the hipGraph (the reference's parity mode switches to the
    upstream dynamic-cache generate, which does not exist here)."""
    t_start = time.time()
    eng, tn, pn, max_frames = _prefill_and_arm(
        talker, talker_input_embeds, attention_mask, trailing_text_hiddens, tts_pad_embed, config, predictor_graph,
        talker_graph, max_new_tokens, min_new_tokens, temperature, top_k, top_p, do_sample, repetition_penalty,
        use_graph=not parity_mode)
    torch.cuda.current_stream(eng.device).synchronize()
    t_prefill = time.time() - t_start
    t_decode_start = time.time()
    issued, n, done = 0, 0, False
    while not done and issued < max_frames:
        issued = run_frames(eng, tn, pn, issued, min(poll_every, max_frames - issued))
        n, done = eng.decode_poll()
    codes = eng.decode_codes(0, n) if n > 0 else None
    torch.cuda.current_stream(eng.device).synchronize()
    t_decode = time.time() - t_decode_start
    timing = {
        "prefill_ms": t_prefill * 1000,
        "decode_s": t_decode,
        "steps": n,
        "ms_per_step": (t_decode / n * 1000) if n > 0 else 0,
        "steps_per_s": (n / t_decode) if t_decode > 0 else 0,
    }
    return codes, timing
