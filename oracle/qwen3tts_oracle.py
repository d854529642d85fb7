"""CPU ORACLE (test infrastructure, NOT the product) for the Qwen3-TTS fast decode path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the shipped path under ``faster-qwen3-tts_amd/`` never does and fails loudly when its HIP
library is missing.

What it restates, in plain eager PyTorch that runs on CPU in fp32 or bf16 (the dtype of the
weights decides, rounding points are those of the module-by-module Torch execution):

* decode-step algebra ...................... reference ``faster_qwen3_tts/generate.py:149-199``
                                            (identical body in ``streaming.py:106-154``)
* predictor 15-codebook schedule ........... reference ``faster_qwen3_tts/predictor_graph.py:115-167``
* sampler / repetition penalty ............. reference ``faster_qwen3_tts/sampling.py:10-66``
* talker KV hand-off, positions, rope delta  reference ``faster_qwen3_tts/talker_graph.py:153-214``
* chunk emission ........................... reference ``faster_qwen3_tts/streaming.py:157-188``
* vocoder call sites and windowing ......... reference ``faster_qwen3_tts/model.py:919-938``, ``:1052-1137``
* layer arithmetic (third party, ``qwen-tts>=0.1.1`` is NOT vendored in the reference and not
  installable here): restated from the architecturally identical sibling shipped in
  transformers 5.15, ``transformers/models/qwen3_omni_moe/modeling_qwen3_omni_moe.py``:
  RMSNorm ``:2229-2243``, q/k-norm GQA attention + rotate_half RoPE ``:2250-2321``, ``:1400-1422``,
  SwiGLU ``:2324-2337``, pre-norm residual layer ``:2340-2379``, RoPE table ``:2401-2436``,
  vocoder blocks ``:3180-3263``, ``:3542-3696``.

Pinning status: the sampler, the decode loop, the streaming chunker and the layer/vocoder blocks
are checked against the reference's own ``sampling.py`` / ``generate.py`` / ``streaming.py`` and the
transformers sibling classes by ``oracle/make_golden.py`` (run in the build container, vectors
committed under ``tests/golden/``) and by ``tests/test_oracle_pins.py``.  The ``qwen-tts`` checkpoint
boundary itself (real weights, tokenizer front-end of the RVQ codec) is **parity unpinned**: no
weights or upstream package exist offline.

Attention numerics: scores, softmax and P.V are evaluated in fp32 over the ``pos+1`` live keys and
rounded once to the activation dtype (the dynamic-cache / SDPA formulation the reference's own
tests treat as ground truth, ``tests/test_e2e_parity.py:177-184``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Weights = Dict[str, torch.Tensor]


# ======================================================================================
# Elementary blocks
# ======================================================================================
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """modeling_qwen3_omni_moe.py:2238-2243."""
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return weight * xf.to(dt)


def rope_inv_freq(head_dim: int, theta: float) -> torch.Tensor:
    """modeling_qwen3_omni_moe.py:2414-2420."""
    return 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))


def rope_cos_sin(positions: torch.Tensor, head_dim: int, theta: float, dtype: torch.dtype):
    """cos/sin rows for 1-D positions (3 equal mRoPE axes collapse to this,
    ``talker_graph.py:210-211``); fp32 angle, cast to activation dtype (``:2436``)."""
    inv = rope_inv_freq(head_dim, theta)
    freqs = positions.to(torch.float32)[:, None] * inv[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x: [n, heads, d]; cos/sin: [n, d] (modeling ``:1418-1422``)."""
    return (x * cos[:, None, :]) + (rotate_half(x) * sin[:, None, :])


def attention_fp32(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: torch.Tensor,
                   scale: float) -> torch.Tensor:
    """q [nq, Hq, d], k/v [nk, Hkv, d], mask [nq, nk] bool (True = attend). GQA by repeat.
    fp32 scores / softmax / PV, one rounding at the end."""
    dt = q.dtype
    Hq, Hkv = q.shape[1], k.shape[1]
    rep = Hq // Hkv
    qf = q.float().permute(1, 0, 2)                      # [Hq, nq, d]
    kf = k.float().permute(1, 0, 2).repeat_interleave(rep, dim=0)
    vf = v.float().permute(1, 0, 2).repeat_interleave(rep, dim=0)
    s = torch.matmul(qf, kf.transpose(1, 2)) * scale     # [Hq, nq, nk]
    s = s.masked_fill(~mask[None, :, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, vf)                              # [Hq, nq, d]
    return o.permute(1, 0, 2).to(dt)


# ======================================================================================
# Transformer stack with a contiguous KV cache
# ======================================================================================
@dataclass
class KVCache:
    k: List[torch.Tensor]
    v: List[torch.Tensor]
    length: int = 0

    @staticmethod
    def empty(n_layers, max_len, n_kv, d, dtype):
        return KVCache([torch.zeros(max_len, n_kv, d, dtype=dtype) for _ in range(n_layers)],
                       [torch.zeros(max_len, n_kv, d, dtype=dtype) for _ in range(n_layers)], 0)


def stack_forward(W: Weights, prefix: str, c, x: torch.Tensor, start: int, cache: KVCache,
                  rope_positions: torch.Tensor, n_pad: int = 0, final_norm: bool = True,
                  window: Optional[int] = None) -> torch.Tensor:
    """Run ``x`` [n, H] (cache slots start..start+n-1) through every layer of a stack.
    Keys ``< n_pad`` are padding (left-padded batch, ``talker_graph.py:172-190``) and masked.
    Returns post-final-norm hidden [n, H]."""
    n = x.shape[0]
    d = c.head_dim
    cos, sin = rope_cos_sin(rope_positions, d, c.rope_theta, x.dtype)
    nk = start + n
    qpos = torch.arange(start, start + n)[:, None]
    kpos = torch.arange(nk)[None, :]
    mask = (kpos <= qpos) & (kpos >= n_pad)
    # left-pad query rows see no real key; let them see themselves so their (discarded, never attended)
    # rows stay finite -- the additive finfo.min mask of the upstream eager path has the same effect
    mask = mask | ((kpos == qpos) & (qpos < n_pad))
    if window is not None:
        mask = mask & (kpos > qpos - window)
    scale = d ** -0.5
    h = x
    for i in range(c.num_hidden_layers):
        p = f"{prefix}.layers.{i}"
        res = h
        hn = rms_norm(h, W[f"{p}.input_layernorm.weight"], c.rms_norm_eps)
        q = F.linear(hn, W[f"{p}.self_attn.q_proj.weight"]).view(n, -1, d)
        k = F.linear(hn, W[f"{p}.self_attn.k_proj.weight"]).view(n, -1, d)
        v = F.linear(hn, W[f"{p}.self_attn.v_proj.weight"]).view(n, -1, d)
        q = rms_norm(q, W[f"{p}.self_attn.q_norm.weight"], c.rms_norm_eps)
        k = rms_norm(k, W[f"{p}.self_attn.k_norm.weight"], c.rms_norm_eps)
        q = apply_rope(q, cos, sin)
        k = apply_rope(k, cos, sin)
        cache.k[i][start:nk] = k
        cache.v[i][start:nk] = v
        a = attention_fp32(q, cache.k[i][:nk], cache.v[i][:nk], mask, scale).reshape(n, -1)
        h = res + F.linear(a, W[f"{p}.self_attn.o_proj.weight"])
        res = h
        hn = rms_norm(h, W[f"{p}.post_attention_layernorm.weight"], c.rms_norm_eps)
        g = F.linear(hn, W[f"{p}.mlp.gate_proj.weight"])
        u = F.linear(hn, W[f"{p}.mlp.up_proj.weight"])
        h = res + F.linear(F.silu(g) * u, W[f"{p}.mlp.down_proj.weight"])
    cache.length = max(cache.length, nk)
    if final_norm:
        h = rms_norm(h, W[f"{prefix}.norm.weight"], c.rms_norm_eps)
    return h


# ======================================================================================
# Sampler (reference sampling.py restated; noise is explicit so GPU and CPU can share it)
# ======================================================================================
def apply_repetition_penalty(logits: torch.Tensor, history: torch.Tensor, penalty: float) -> torch.Tensor:
    """sampling.py:10-29 (in place)."""
    if penalty == 1.0 or history.numel() == 0:
        return logits
    u = history.unique()
    t = logits[..., u]
    logits[..., u] = torch.where(t > 0, t / penalty, t * penalty)
    return logits


def sample_logits(logits: torch.Tensor, *, temperature: float, top_k: int, top_p: float, do_sample: bool,
                  suppress_mask: Optional[torch.Tensor] = None, suppress_tokens=None,
                  noise: Optional[torch.Tensor] = None, stable_top_p: bool = True) -> torch.Tensor:
    """sampling.py:32-66.  ``noise`` = Exp(1) variates in the logits dtype, one per vocabulary entry:
    ``torch.multinomial(p, 1)`` is ``argmax(p / q)`` with ``q = empty_like(p).exponential_(1)``
    (ATen ``multinomial`` single-sample fast path), so drawing ``q`` with the same generator state
    reproduces the reference call bit for bit.  ``stable_top_p``: the reference's top-p branch uses an
    UNSTABLE ``torch.sort`` (sampling.py:58), so which of several exactly-tied logits survive a nucleus
    cut that falls inside the tie group is implementation-defined there; the HIP path and this oracle
    fix it to lowest-index-first (stable).  ``stable_top_p=False`` calls the reference's exact op."""
    logits = logits.clone()
    if suppress_mask is not None:
        logits[..., suppress_mask] = float("-inf")
    if suppress_tokens:
        logits[..., list(suppress_tokens)] = float("-inf")
    if not do_sample:
        return torch.argmax(logits, dim=-1)
    logits = logits / temperature
    if top_k > 0:
        kth = torch.topk(logits, min(top_k, logits.size(-1)))[0][..., -1:]
        logits = torch.where(logits < kth, torch.full_like(logits, float("-inf")), logits)
    if top_p < 1.0:
        sl, si = torch.sort(logits, descending=True, stable=True) if stable_top_p else torch.sort(logits, descending=True)
        pr = F.softmax(sl, dim=-1)
        cum = torch.cumsum(pr, dim=-1)
        rm = cum > top_p
        rm[..., 0] = False
        sl[rm] = float("-inf")
        logits = torch.full_like(logits, float("-inf"))
        logits.scatter_(-1, si, sl)
    probs = F.softmax(logits, dim=-1)
    if noise is None:
        noise = torch.empty_like(probs).exponential_(1)
    return torch.argmax(probs / noise.to(probs.dtype).view_as(probs), dim=-1)


def build_suppress_mask(vocab: int, eos_id: int) -> torch.Tensor:
    """generate.py:46-50."""
    m = torch.zeros(vocab, dtype=torch.bool)
    m[max(0, vocab - 1024):] = True
    m[eos_id] = False
    return m


# ======================================================================================
# Model-level oracle
# ======================================================================================
@dataclass
class SamplingParams:
    temperature: float = 0.9
    top_k: int = 50
    top_p: float = 1.0
    do_sample: bool = True
    repetition_penalty: float = 1.05
    min_new_tokens: int = 2
    max_new_tokens: int = 2048


GREEDY = dict(temperature=1.0, top_k=0, top_p=1.0, do_sample=False, repetition_penalty=1.0, min_new_tokens=0)


class OracleTTS:
    """Eager Torch restatement of talker + predictor + decode loop over a weight table."""

    def __init__(self, cfg, W: Weights, max_seq_len: int = 2048):
        self.cfg, self.W = cfg, W
        self.max_seq_len = max_seq_len
        self.dtype = W["talker.codec_head.weight"].dtype
        t, p = cfg.talker, cfg.predictor
        self.tcache = KVCache.empty(t.num_hidden_layers, max_seq_len, t.num_key_value_heads, t.head_dim, self.dtype)
        self.pcache = KVCache.empty(p.num_hidden_layers, 2 + cfg.num_code_groups - 1, p.num_key_value_heads,
                                    p.head_dim, self.dtype)
        self.n_pad = 0
        self.rope_delta = 0.0
        # predictor sampling policy is construction-time state (model.py:209-218)
        self.pred_sampling = dict(do_sample=True, top_k=50, top_p=1.0, temperature=0.9)
        self.margins: List[float] = []        # top-2 logit gap of every first-codebook decision
        self.pred_margins: List[float] = []   # same for every predictor decision (15 per frame)
        self.top1: List[float] = []           # winning logit of every first-codebook decision (scale for ulp attribution)
        self.pred_top1: List[float] = []

    # ---- talker ----------------------------------------------------------------------
    def prefill(self, embeds: torch.Tensor, attention_mask: torch.Tensor):
        """generate.py:107-122: returns (logits[V], past_hidden[1,1,H], gen_step, prefill_len)."""
        x = embeds[0].to(self.dtype)
        L = x.shape[0]
        if L > self.max_seq_len:
            raise RuntimeError(f"Input is too long: prefill has {L} tokens but max_seq_len={self.max_seq_len}. "
                               "Use shorter text or shorter reference audio.")
        self.n_pad = int((attention_mask[0] == 0).sum())
        self.rope_delta = float(-self.n_pad)
        self.tcache.length = 0
        pos = (torch.arange(L) + self.rope_delta).clamp(min=0) if self.n_pad else torch.arange(L).float()
        h = stack_forward(self.W, "talker.model", self.cfg.talker, x, 0, self.tcache, pos, n_pad=self.n_pad)
        logits = F.linear(h[-1], self.W["talker.codec_head.weight"])
        return logits, h[-1].view(1, 1, -1), 0, L

    def talker_step(self, embeds: torch.Tensor, position: int) -> torch.Tensor:
        """talker_graph.py:198-214: one token at cache slot ``position``; RoPE position =
        position + rope_delta; returns post-norm hidden [1,1,H]."""
        x = embeds.reshape(1, -1).to(self.dtype)
        rp = torch.tensor([position + self.rope_delta], dtype=torch.float32)
        h = stack_forward(self.W, "talker.model", self.cfg.talker, x, position, self.tcache, rp, n_pad=self.n_pad)
        return h.view(1, 1, -1)

    def codec_head(self, hidden: torch.Tensor) -> torch.Tensor:
        return F.linear(hidden.reshape(-1), self.W["talker.codec_head.weight"])

    # ---- predictor -------------------------------------------------------------------
    def _proj(self, x):
        w = self.W.get("talker.code_predictor.small_to_mtp_projection.weight")
        if w is None:
            return x
        return F.linear(x, w, self.W.get("talker.code_predictor.small_to_mtp_projection.bias"))

    def predictor_loop(self, pred_input: torch.Tensor, noise: Optional[torch.Tensor] = None,
                       return_logits: bool = False):
        """predictor_graph.py:115-167.  pred_input [1,2,H] = cat(past_hidden, embed(tok0)).
        noise: [15, Vp] Exp(1) rows or None.  Returns LongTensor[15]."""
        cfg = self.cfg
        pre = "talker.code_predictor"
        nc = cfg.num_code_groups - 1
        self.pcache.length = 0            # static_cache.reset(), predictor_graph.py:212
        h = self._proj(pred_input[0].to(self.dtype))                      # [2, Hp]
        h = stack_forward(self.W, f"{pre}.model", cfg.predictor, h, 0, self.pcache, torch.arange(2).float())
        toks, all_logits = [], []
        logits = F.linear(h[-1:], self.W[f"{pre}.lm_head.0.weight"])       # [1, Vp]
        all_logits.append(logits[0])
        self.pred_margins.append(self._margin(logits)); self.pred_top1.append(float(logits.float().max()))
        tok = sample_logits(logits, noise=None if noise is None else noise[0], **self.pred_sampling)
        toks.append(tok[0])
        for cb in range(1, nc):
            emb = F.embedding(tok, self.W[f"{pre}.model.codec_embedding.{cb - 1}.weight"])   # [1, H]
            emb = self._proj(emb)
            h = stack_forward(self.W, f"{pre}.model", cfg.predictor, emb, 1 + cb, self.pcache,
                              torch.tensor([1.0 + cb]))
            logits = F.linear(h[-1:], self.W[f"{pre}.lm_head.{cb}.weight"])
            all_logits.append(logits[0])
            self.pred_margins.append(self._margin(logits)); self.pred_top1.append(float(logits.float().max()))
            tok = sample_logits(logits, noise=None if noise is None else noise[cb], **self.pred_sampling)
            toks.append(tok[0])
        out = torch.stack(toks).to(torch.long)
        return (out, torch.stack(all_logits)) if return_logits else out

    # ---- decode loop -----------------------------------------------------------------
    def embed_frame(self, token: torch.Tensor, codes15: torch.Tensor, text_add: torch.Tensor):
        """generate.py:154,162-171: 16-way embedding sum + text/pad embed -> [1,1,H]."""
        W = self.W
        hs = [F.embedding(token.view(1, 1), W["talker.model.codec_embedding.weight"])]
        for i in range(self.cfg.num_code_groups - 1):
            hs.append(F.embedding(codes15[i].view(1, 1), W[f"talker.code_predictor.model.codec_embedding.{i}.weight"]))
        e = torch.cat(hs, dim=1).sum(1, keepdim=True)
        return e + text_add

    def generate(self, tie, tam, tth, tpe, sp: SamplingParams, talker_noise: Optional[torch.Tensor] = None,
                 pred_noise: Optional[torch.Tensor] = None, record_margins: bool = False):
        """generate.py:99-215 (fast path).  talker_noise [max_new+1, V], pred_noise [max_new, 15, Vp]
        are optional pre-drawn Exp(1) variates (row 0 of talker_noise feeds the prefill sample).
        Returns LongTensor[T,16] or None."""
        cfg = self.cfg
        eos = cfg.codec_eos_token_id
        V = cfg.talker.vocab_size
        sm = build_suppress_mask(V, eos)
        kw = dict(temperature=sp.temperature, top_k=sp.top_k, top_p=sp.top_p, do_sample=sp.do_sample)
        logits, past_hidden, gen_step, prefill_len = self.prefill(tie, tam)
        self.margins, self.pred_margins, self.top1, self.pred_top1 = [], [], [], []
        if record_margins:
            self._record_margin(logits, sm, [eos] if sp.min_new_tokens > 0 else None)
        token = sample_logits(logits.view(1, -1), suppress_mask=sm,
                              suppress_tokens=[eos] if sp.min_new_tokens > 0 else None,
                              noise=None if talker_noise is None else talker_noise[0], **kw)
        codes: List[torch.Tensor] = []
        tth_d, tpe_d = tth.to(self.dtype), tpe.to(self.dtype)
        for step in range(sp.max_new_tokens):
            if int(token) == eos:
                break
            last = F.embedding(token.view(1, 1), self.W["talker.model.codec_embedding.weight"])
            pred_in = torch.cat((past_hidden, last), dim=1)
            c15 = self.predictor_loop(pred_in, None if pred_noise is None else pred_noise[step])
            codes.append(torch.cat([token.view(1), c15]))
            text_add = tth_d[:, gen_step].unsqueeze(1) if gen_step < tth_d.shape[1] else tpe_d
            x = self.embed_frame(token, c15, text_add)
            pos = prefill_len + step
            if pos >= self.max_seq_len - 1:
                break
            hidden = self.talker_step(x, pos)
            logits = self.codec_head(hidden).view(1, 1, -1)
            if sp.repetition_penalty != 1.0 and codes:
                hist = torch.stack([c[0] for c in codes])
                logits = apply_repetition_penalty(logits, hist, sp.repetition_penalty)
            sup = [eos] if len(codes) < sp.min_new_tokens else None
            if record_margins:
                self._record_margin(logits.view(-1), sm, sup)
            token = sample_logits(logits.squeeze(0), suppress_mask=sm, suppress_tokens=sup,
                                  noise=None if talker_noise is None else talker_noise[step + 1], **kw)
            past_hidden = hidden.clone()
            gen_step += 1
        return torch.stack(codes) if codes else None

    @staticmethod
    def _margin(logits) -> float:
        t = torch.topk(logits.detach().float().view(-1), 2)[0]
        return float(t[0] - t[1])

    def _record_margin(self, logits, sm, sup):
        l = logits.detach().float().view(-1).clone()
        l[sm] = float("-inf")
        if sup:
            l[list(sup)] = float("-inf")
        top2 = torch.topk(l, 2)[0]
        self.margins.append(float(top2[0] - top2[1]))
        self.top1.append(float(top2[0]))


def stream_chunks(codes: Optional[torch.Tensor], chunk_size: int):
    """streaming.py:157-188: the chunk boundaries / flags a streaming run must reproduce."""
    if codes is None:
        return
    T = codes.shape[0]
    n_chunks = (T + chunk_size - 1) // chunk_size
    total = 0
    for i in range(n_chunks):
        c = codes[i * chunk_size:(i + 1) * chunk_size]
        total += c.shape[0]
        # final flag: True only for a trailing partial chunk or when generation stopped exactly here
        yield c, dict(chunk_index=i, chunk_steps=c.shape[0], total_steps_so_far=total)


# ======================================================================================
# 12 Hz codec decoder (vocoder)
# ======================================================================================
def causal_conv1d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], dilation: int = 1,
                  groups: int = 1) -> torch.Tensor:
    """modeling ``:3180-3217`` with stride 1: left pad (k-1)*dilation zeros.  x [C, T]."""
    k = (w.shape[-1] - 1) * dilation + 1
    xp = F.pad(x.unsqueeze(0), (k - 1, 0))
    return F.conv1d(xp, w, b, dilation=dilation, groups=groups)[0]


def causal_trans_conv1d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], stride: int) -> torch.Tensor:
    """modeling ``:3220-3233``: ConvTranspose1d then trim ``k - stride`` samples on BOTH sides
    (the sibling code trims ``left_pad == right_pad``), so k=2s yields (T-1)*s samples."""
    k = w.shape[-1]
    y = F.conv_transpose1d(x.unsqueeze(0), w, b, stride=stride)[0]
    pad = k - stride
    return y[..., pad: y.shape[-1] - pad]


def snake_beta(x: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """modeling ``:3566-3580``; x [C, T]."""
    a = torch.exp(alpha)[:, None]
    b = torch.exp(beta)[:, None]
    return x + (1.0 / (b + 1e-9)) * torch.pow(torch.sin(x * a), 2)


def convnext_block(x, W, p):
    """modeling ``:3236-3263``; x [C, T]."""
    inp = x
    h = causal_conv1d(x, W[f"{p}.dwconv.conv.weight"], W[f"{p}.dwconv.conv.bias"], groups=x.shape[0])
    h = h.transpose(0, 1)
    h = F.layer_norm(h, (h.shape[-1],), W[f"{p}.norm.weight"], W[f"{p}.norm.bias"], 1e-6)
    h = F.linear(h, W[f"{p}.pwconv1.weight"], W[f"{p}.pwconv1.bias"])
    h = F.gelu(h)
    h = F.linear(h, W[f"{p}.pwconv2.weight"], W[f"{p}.pwconv2.bias"])
    h = W[f"{p}.gamma"] * h
    return inp + h.transpose(0, 1)


def codec_transformer_core(h: torch.Tensor, W: Weights, c, t: str = "decoder.pre_transformer") -> torch.Tensor:
    """Sliding-window pre-norm transformer with layer scale and final RMSNorm
    (modeling ``:3372-3540``).  h [T, hidden] -> [T, hidden]."""
    T = h.shape[0]
    d = c.head_dim
    cos, sin = rope_cos_sin(torch.arange(T).float(), d, c.rope_theta, h.dtype)
    qpos = torch.arange(T)[:, None]
    kpos = torch.arange(T)[None, :]
    mask = (kpos <= qpos) & (kpos > qpos - c.sliding_window)
    for i in range(c.num_hidden_layers):
        p = f"{t}.layers.{i}"
        res = h
        hn = rms_norm(h, W[f"{p}.input_layernorm.weight"], c.rms_norm_eps)
        q = F.linear(hn, W[f"{p}.self_attn.q_proj.weight"]).view(T, -1, d)
        k = F.linear(hn, W[f"{p}.self_attn.k_proj.weight"]).view(T, -1, d)
        v = F.linear(hn, W[f"{p}.self_attn.v_proj.weight"]).view(T, -1, d)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        a = attention_fp32(q, k, v, mask, d ** -0.5).reshape(T, -1)
        h = res + W[f"{p}.self_attn_layer_scale.scale"] * F.linear(a, W[f"{p}.self_attn.o_proj.weight"])
        res = h
        hn = rms_norm(h, W[f"{p}.post_attention_layernorm.weight"], c.rms_norm_eps)
        m = F.linear(F.silu(F.linear(hn, W[f"{p}.mlp.gate_proj.weight"])) * F.linear(hn, W[f"{p}.mlp.up_proj.weight"]),
                     W[f"{p}.mlp.down_proj.weight"])
        h = res + W[f"{p}.mlp_layer_scale.scale"] * m
    return rms_norm(h, W[f"{t}.norm.weight"], c.rms_norm_eps)


def codec_transformer(x: torch.Tensor, W: Weights, c) -> torch.Tensor:
    """``input_proj`` -> transformer core -> ``output_proj``.  x [T, latent] -> [T, latent]."""
    t = "decoder.pre_transformer"
    h = F.linear(x, W[f"{t}.input_proj.weight"], W[f"{t}.input_proj.bias"])
    h = codec_transformer_core(h, W, c, t)
    return F.linear(h, W[f"{t}.output_proj.weight"], W[f"{t}.output_proj.bias"])


def decoder_block(h: torch.Tensor, W: Weights, b: str, r: int) -> torch.Tensor:
    """modeling ``:3610-3633``: SnakeBeta, causal ConvTranspose(k=2r, s=r), 3 residual units (d=1,3,9)."""
    h = snake_beta(h, W[f"{b}.0.alpha"], W[f"{b}.0.beta"])
    h = causal_trans_conv1d(h, W[f"{b}.1.conv.weight"], W[f"{b}.1.conv.bias"], r)
    for j, dil in enumerate((1, 3, 9)):
        u = f"{b}.{j + 2}"
        res = h
        y = snake_beta(h, W[f"{u}.act1.alpha"], W[f"{u}.act1.beta"])
        y = causal_conv1d(y, W[f"{u}.conv1.conv.weight"], W[f"{u}.conv1.conv.bias"], dilation=dil)
        y = snake_beta(y, W[f"{u}.act2.alpha"], W[f"{u}.act2.beta"])
        y = causal_conv1d(y, W[f"{u}.conv2.conv.weight"], W[f"{u}.conv2.conv.bias"])
        h = y + res
    return h


def rvq_decode(codes: torch.Tensor, W: Weights, c) -> torch.Tensor:
    """Split RVQ: semantic codebook(s) and acoustic codebooks each sum their rows and go through
    their own 1x1 ``output_proj``; the two are added.  codes [T, nq] -> [codebook_dim, T]."""
    p = "decoder.quantizer"
    ns = c.num_semantic_quantizers
    out = None
    for name, lo, hi in (("rvq_first", 0, ns), ("rvq_rest", ns, c.num_quantizers)):
        acc = None
        for j in range(lo, hi):
            e = F.embedding(codes[:, j], W[f"{p}.{name}.vq.layers.{j - lo}._codebook.embedding"])
            acc = e if acc is None else acc + e
        y = F.conv1d(acc.transpose(0, 1).unsqueeze(0), W[f"{p}.{name}.output_proj.weight"])[0]
        out = y if out is None else out + y
    return out


def codec_decode(codes: torch.Tensor, W: Weights, c) -> torch.Tensor:
    """codes [T, 16] int64 -> waveform [n_samples] (modeling ``:3675-3686`` with the TTS tokenizer's
    RVQ front-end and ``pre_conv``)."""
    p = "decoder"
    h = rvq_decode(codes, W, c)                                                   # [512, T]
    h = causal_conv1d(h, W[f"{p}.pre_conv.conv.weight"], W[f"{p}.pre_conv.conv.bias"])  # [latent, T]
    h = codec_transformer(h.transpose(0, 1), W, c).transpose(0, 1)
    for i, f in enumerate(c.upsampling_ratios):
        h = causal_trans_conv1d(h, W[f"{p}.upsample.{i}.0.conv.weight"], W[f"{p}.upsample.{i}.0.conv.bias"], f)
        h = convnext_block(h, W, f"{p}.upsample.{i}.1")
    d = f"{p}.decoder"
    h = causal_conv1d(h, W[f"{d}.0.conv.weight"], W[f"{d}.0.conv.bias"])
    for i, r in enumerate(c.upsample_rates):
        h = decoder_block(h, W, f"{d}.{i + 1}.block", r)
    n = len(c.upsample_rates)
    h = snake_beta(h, W[f"{d}.{n + 1}.alpha"], W[f"{d}.{n + 1}.beta"])
    h = causal_conv1d(h, W[f"{d}.{n + 2}.conv.weight"], W[f"{d}.{n + 2}.conv.bias"])
    return h.clamp(min=-1, max=1).reshape(-1)


def codec_chunked_decode(codes: torch.Tensor, W: Weights, c, chunk_size: int = 300, left_context_size: int = 25) -> torch.Tensor:
    """modeling ``:3686-3696`` (``chunked_decode``): long inputs are decoded in ``chunk_size``-frame pieces, each with up
    to ``left_context_size`` frames of left context whose ``context * total_upsample`` samples are dropped.  One piece
    (T <= chunk_size) is the plain decode."""
    T = codes.shape[0]
    wavs, start = [], 0
    while start < T:
        end = min(start + chunk_size, T)
        ctx = left_context_size if start - left_context_size > 0 else start
        w = codec_decode(codes[start - ctx:end], W, c)
        wavs.append(w[ctx * c.total_upsample:])
        start = end
    return torch.cat(wavs)


class OracleSpeechTokenizer:
    """Duck type of upstream ``speech_tokenizer`` as the reference calls it
    (``model.py:924``; payload shape pinned by reference ``tests/test_sample_rate.py:53-75``)."""

    def __init__(self, cfg, W):
        self.cfg, self.W = cfg, W
        self.sample_rate = cfg.codec.sample_rate

    def decode(self, payload):
        codes = payload["audio_codes"]
        return [codec_chunked_decode(codes[b], self.W, self.cfg.codec) for b in range(codes.shape[0])], self.sample_rate


def streaming_vocode(tok, chunks, ref_codes, chunk_size, context_frames: int = 25):
    """model.py:1052-1137 restated: phase-1 accumulated decode with calibration, phase-2 25-frame
    left-context sliding window.  Yields the new-audio arrays."""
    min_cal = max(context_frames, chunk_size)
    all_codes, prev_len, spf = [], 0, None
    for chunk in chunks:
        all_codes.append(chunk)
        n_new = chunk.shape[0]
        flat = torch.cat(all_codes, 0)
        n_total = flat.shape[0]
        if spf is None:
            inp = torch.cat([ref_codes, flat], 0) if ref_codes is not None else flat
            audio = tok.decode({"audio_codes": inp.unsqueeze(0)})[0][0].flatten().float().numpy()
            if ref_codes is not None:
                cut = int(ref_codes.shape[0] / max(inp.shape[0], 1) * len(audio))
                gen = audio[cut:]
            else:
                gen = audio
            new = gen[prev_len:]
            prev_len = len(gen)
            if n_total >= min_cal:
                spf = len(gen) / n_total
        else:
            start = max(0, n_total - n_new - context_frames)
            win = flat[start:]
            n_ctx = win.shape[0] - n_new
            audio = tok.decode({"audio_codes": win.unsqueeze(0)})[0][0].flatten().float().numpy()
            new = audio[int(round(n_ctx * spf)):] if n_ctx > 0 else audio
        yield new


# ======================================================================================
# Prompt side: a CPU duck type of the upstream model object the reference's prompt builder reaches into
# ======================================================================================
class OraclePromptModel:
    """``m`` as ``FasterQwen3TTS._build_talker_inputs_local(self, m, ...)`` uses it (reference ``model.py:583-805``):
    ``m.talker.{device, get_input_embeddings, get_text_embeddings, text_projection}``, ``m.config`` (+ ``talker_config``),
    ``m.generate_speaker_prompt`` and ``m.generate_icl_prompt``.  The MLP is the sibling's ResizeMLP
    (``modeling_qwen3_omni_moe.py:2207-2215``: fc1, SiLU, fc2).  ``generate_icl_prompt`` restates upstream behaviour
    [recalled: qwen-tts is not vendored] -- parity unpinned for that one function."""

    def __init__(self, cfg, W: Weights):
        from types import SimpleNamespace as NS
        self.W, self._cfg = W, cfg
        emb = lambda w: (lambda ids: F.embedding(ids, w))
        self.talker = NS(
            device=torch.device("cpu"),
            get_input_embeddings=lambda: emb(W["talker.model.codec_embedding.weight"]),
            get_text_embeddings=lambda: emb(W["talker.model.text_embedding.weight"]),
            text_projection=lambda x: F.linear(F.silu(F.linear(x, W["talker.text_projection.linear_fc1.weight"],
                                                                 W["talker.text_projection.linear_fc1.bias"])),
                                               W["talker.text_projection.linear_fc2.weight"], W["talker.text_projection.linear_fc2.bias"]),
            code_predictor=NS(get_input_embeddings=lambda: [emb(W[f"talker.code_predictor.model.codec_embedding.{j}.weight"])
                                                            for j in range(cfg.num_code_groups - 1)]))
        tc = NS(codec_eos_token_id=cfg.codec_eos_token_id, codec_pad_id=cfg.codec_pad_id, codec_bos_id=cfg.codec_bos_id,
                codec_think_id=cfg.codec_think_id, codec_nothink_id=cfg.codec_nothink_id, codec_think_bos_id=cfg.codec_think_bos_id,
                codec_think_eos_id=cfg.codec_think_eos_id, codec_language_id=dict(cfg.codec_language_id), spk_id=dict(cfg.spk_id),
                spk_is_dialect=dict(cfg.spk_is_dialect), vocab_size=cfg.talker.vocab_size, num_code_groups=cfg.num_code_groups)
        self.config = NS(talker_config=tc, tts_bos_token_id=cfg.tts_bos_token_id, tts_eos_token_id=cfg.tts_eos_token_id,
                         tts_pad_token_id=cfg.tts_pad_token_id)

    def generate_speaker_prompt(self, voice_clone_prompt):
        dt = self.W["talker.codec_head.weight"].dtype
        return [torch.as_tensor(e).to(dt) for e in voice_clone_prompt["ref_spk_embedding"]]

    def generate_icl_prompt(self, text_id, ref_id, ref_code, tts_pad_embed, tts_eos_embed, non_streaming_mode):
        t, cfg = self.talker, self._cfg
        text_embed = t.text_projection(t.get_text_embeddings()(torch.cat([ref_id, text_id], dim=-1)))
        text_embed = torch.cat([text_embed, tts_eos_embed], dim=1)
        embs = [t.get_input_embeddings()(ref_code[:, :1])]
        pe = t.code_predictor.get_input_embeddings()
        for i in range(1, cfg.num_code_groups):
            embs.append(pe[i - 1](ref_code[:, i:i + 1]))
        codec_embed = torch.cat(embs, dim=1).sum(1).unsqueeze(0)
        bos = t.get_input_embeddings()(torch.tensor([[cfg.codec_bos_id]]))
        codec_embed = torch.cat([bos, codec_embed], dim=1)
        tl, cl = text_embed.shape[1], codec_embed.shape[1]
        if non_streaming_mode:
            pad = t.get_input_embeddings()(torch.tensor([[cfg.codec_pad_id] * tl]))
            return torch.cat([text_embed + pad, codec_embed + tts_pad_embed], dim=1), tts_pad_embed
        if tl > cl:
            return text_embed[:, :cl] + codec_embed, text_embed[:, cl:]
        text_embed = torch.cat([text_embed] + [tts_pad_embed] * (cl - tl), dim=1)
        return text_embed + codec_embed, tts_pad_embed
