"""Host-side owner of one ``fq3_ctx``: weight packing, RoPE tables, and typed wrappers over the C ABI.

PyTorch is plumbing here (device memory for weights / activations, the stream handle); every
arithmetic step of the decode path runs in ``libfq3hip.so``.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, Optional

import torch

from . import _lib as L
from .config import TTSConfig, StackConfig

Weights = Dict[str, torch.Tensor]
_CAPTURE_LOCK = threading.Lock()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def rope_tables(head_dim: int, theta: float, n_pos: int, dtype: torch.dtype):
    """cos/sin [n_pos, head_dim/2] as fp32 images of the activation-dtype values, computed with the
    same Torch ops as the upstream rotary module (inv_freq in fp32, fp32 angle, cast to dtype) so the
    kernels see bit-identical tables.  (transformers modeling_qwen3_omni_moe.py:2414-2436.)"""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return ang.cos().to(dtype).float().contiguous(), ang.sin().to(dtype).float().contiguous()


def _fill_config(c, cfg: TTSConfig, dtype: torch.dtype, max_seq_len: int, max_frames: int, has_projection: bool):
    c.dtype = L.FQ3_BF16 if dtype == torch.bfloat16 else L.FQ3_F32
    for dst, src in ((c.talker, cfg.talker), (c.predictor, cfg.predictor)):
        dst.hidden, dst.inter, dst.n_layers = src.hidden_size, src.intermediate_size, src.num_hidden_layers
        dst.n_heads, dst.n_kv_heads, dst.head_dim = src.num_attention_heads, src.num_key_value_heads, src.head_dim
        dst.vocab, dst.rms_eps = src.vocab_size, src.rms_norm_eps
    c.num_code_groups = cfg.num_code_groups
    c.max_seq_len = int(max_seq_len)
    c.codec_eos_token_id = cfg.codec_eos_token_id
    c.has_projection = 1 if has_projection else 0
    c.max_frames = int(max_frames)
    return c


KV_BLOCK = 64           # keys per block of the paged talker cache (kKeysPerTile of csrc/decode_kernels.cuh)


class Fq3KvPool:
    """A pool of 64-key KV blocks shared by the decode contexts of one scheduler (``fq3_kv_pool_*``): a context built with
    ``Fq3Engine(..., pool=pool)`` owns nothing while idle, its prompt's blocks after a prefill and those of
    ``prefill_len + max_new_tokens`` once armed; a lane takes a staged prompt over by exchanging block ids
    (``Fq3Engine.kv_adopt``) instead of copying rows.  ``Fq3Error`` with code ``FQ3_ENOMEM`` when the pool runs short."""

    def __init__(self, cfg: TTSConfig, n_blocks: int, device: str = "cuda", dtype: torch.dtype = torch.bfloat16):
        self.lib = L.load()
        self.device = torch.device(device)
        self.dtype = dtype
        self.handle = L.vp()
        c = _fill_config(L.Config(), cfg, dtype, 64, 8, False)
        with torch.cuda.device(self.device):
            L.check(self.lib.fq3_kv_pool_create(C.byref(c), int(n_blocks), C.byref(self.handle)))

    def stats(self) -> Dict[str, int]:
        n, f, h, b = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int64(0)
        L.check(self.lib.fq3_kv_pool_stats(self.handle, C.byref(n), C.byref(f), C.byref(h), C.byref(b)))
        return {"blocks": n.value, "free": f.value, "high_water": h.value, "bytes_per_block": b.value}

    @staticmethod
    def blocks_for(n_positions: int) -> int:
        return (int(n_positions) + KV_BLOCK - 1) // KV_BLOCK

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            if self.lib.fq3_kv_pool_destroy(self.handle) == 0:      # refuses (and stays alive) while contexts are attached
                self.handle = L.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Fq3Engine:
    """One decode context (talker + predictor static state) on one GPU.  Not re-entrant."""

    def __init__(self, cfg: TTSConfig, weights: Weights, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                 max_seq_len: int = 2048, max_frames: int = 4096, share: Optional["Fq3Engine"] = None,
                 pool: Optional[Fq3KvPool] = None):
        """``share``: another engine on the same device/dtype whose packed weight tensors this context borrows
        (one weight replica per GPU, one context -- KV cache, loop state, graph -- per concurrent utterance).
        ``pool``: draw the talker's KV blocks from this shared pool (``fq3_ctx_create_pooled``) instead of reserving
        ``max_seq_len`` slots privately."""
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("fq3hip supports torch.bfloat16 and torch.float32")
        self.lib = L.load()
        self.cfg, self.dtype, self.max_seq_len = cfg, dtype, int(max_seq_len)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("fq3hip needs a ROCm GPU device ('cuda' in PyTorch-ROCm)")
        self.max_frames = int(max_frames)
        self._keep = []          # tensors whose storage the context borrows
        self.ctx = L.vp()
        self.pool = pool         # keeps the pool alive as long as this context
        c = _fill_config(L.Config(), cfg, dtype, self.max_seq_len, self.max_frames,
                         "talker.code_predictor.small_to_mtp_projection.weight" in weights or
                         (share is not None and bool(share._table.proj_w)))
        with torch.cuda.device(self.device):
            if pool is not None:
                L.check(self.lib.fq3_ctx_create_pooled(C.byref(c), pool.handle, C.byref(self.ctx)))
            else:
                L.check(self.lib.fq3_ctx_create(C.byref(c), C.byref(self.ctx)))
            if share is not None:
                if share.dtype != dtype or share.device != self.device or share.max_seq_len < self.max_seq_len:
                    raise ValueError("shared engine must match dtype/device and cover max_seq_len (RoPE tables)")
                self._share = share                       # keeps the borrowed tensors alive
                self._table = share._table
                self.codec_embedding, self.codec_head_w = share.codec_embedding, share.codec_head_w
                self.predictor_embeddings = share.predictor_embeddings
                L.check(self.lib.fq3_bind_weights(self.ctx, C.byref(self._table)))
            else:
                self._bind(weights)
        self.weights = weights
        self._pred_sampling = dict(do_sample=True, top_k=50, top_p=1.0, temperature=0.9)

    @classmethod
    def sampler_only(cls, device, dtype: torch.dtype) -> "Fq3Engine":
        """Weight-less context (minimal dims) for callers that only need ``sample`` (fq3hip.sampling)."""
        from .config import StackConfig
        self = cls.__new__(cls)
        self.lib = L.load()
        self.cfg = TTSConfig(talker=StackConfig(hidden_size=64, intermediate_size=64, num_hidden_layers=1,
                                                num_attention_heads=1, num_key_value_heads=1, vocab_size=4096),
                             predictor=StackConfig(hidden_size=64, intermediate_size=64, num_hidden_layers=1,
                                                   num_attention_heads=1, num_key_value_heads=1, vocab_size=4096),
                             codec_eos_token_id=-1)
        self.dtype, self.max_seq_len, self.max_frames = dtype, 8, 8
        self.device = torch.device(device)
        self._keep, self.weights = [], {}
        self.ctx = L.vp()
        c = L.Config()
        c.dtype = L.FQ3_BF16 if dtype == torch.bfloat16 else L.FQ3_F32
        for dst in (c.talker, c.predictor):
            dst.hidden, dst.inter, dst.n_layers, dst.n_heads, dst.n_kv_heads, dst.head_dim = 64, 64, 1, 1, 1, 128
            dst.vocab, dst.rms_eps = 4096, 1e-6
        c.num_code_groups, c.max_seq_len, c.codec_eos_token_id, c.has_projection, c.max_frames = 16, 8, -1, 0, 8
        with torch.cuda.device(self.device):
            L.check(self.lib.fq3_ctx_create(C.byref(c), C.byref(self.ctx)))
        return self

    # ------------------------------------------------------------------------------------------
    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        t = t.to(device=self.device, dtype=self.dtype).contiguous()
        self._keep.append(t)
        return t

    def _layers(self, W: Weights, prefix: str, sc: StackConfig):
        arr = (L.LayerWeights * sc.num_hidden_layers)()
        for i in range(sc.num_hidden_layers):
            p = f"{prefix}.layers.{i}"
            qkv = self._dev(torch.cat([W[f"{p}.self_attn.q_proj.weight"], W[f"{p}.self_attn.k_proj.weight"],
                                       W[f"{p}.self_attn.v_proj.weight"]], 0))
            gu = self._dev(torch.cat([W[f"{p}.mlp.gate_proj.weight"], W[f"{p}.mlp.up_proj.weight"]], 0))
            lw = arr[i]
            lw.input_norm = _ptr(self._dev(W[f"{p}.input_layernorm.weight"]))
            lw.qkv = _ptr(qkv)
            lw.q_norm = _ptr(self._dev(W[f"{p}.self_attn.q_norm.weight"]))
            lw.k_norm = _ptr(self._dev(W[f"{p}.self_attn.k_norm.weight"]))
            lw.o = _ptr(self._dev(W[f"{p}.self_attn.o_proj.weight"]))
            lw.post_norm = _ptr(self._dev(W[f"{p}.post_attention_layernorm.weight"]))
            lw.gate_up = _ptr(gu)
            lw.down = _ptr(self._dev(W[f"{p}.mlp.down_proj.weight"]))
        return arr

    def _bind(self, W: Weights):
        cfg = self.cfg
        t = L.WeightTable()
        self._tl = self._layers(W, "talker.model", cfg.talker)
        self._pl = self._layers(W, "talker.code_predictor.model", cfg.predictor)
        t.talker_layers, t.predictor_layers = self._tl, self._pl
        self.codec_embedding = self._dev(W["talker.model.codec_embedding.weight"])
        self.codec_head_w = self._dev(W["talker.codec_head.weight"])
        t.talker_final_norm = _ptr(self._dev(W["talker.model.norm.weight"]))
        t.codec_embedding = _ptr(self.codec_embedding)
        t.codec_head = _ptr(self.codec_head_w)
        pre = "talker.code_predictor"
        t.predictor_final_norm = _ptr(self._dev(W[f"{pre}.model.norm.weight"]))
        pw = W.get(f"{pre}.small_to_mtp_projection.weight")
        pb = W.get(f"{pre}.small_to_mtp_projection.bias")
        t.proj_w = _ptr(self._dev(pw)) if pw is not None else None
        t.proj_b = _ptr(self._dev(pb)) if pb is not None else None
        n = cfg.num_code_groups - 1
        self.predictor_embeddings = [self._dev(W[f"{pre}.model.codec_embedding.{j}.weight"]) for j in range(n)]
        self._pe = (L.vp * n)(*[e.data_ptr() for e in self.predictor_embeddings])
        self._lh = (L.vp * n)(*[self._dev(W[f"{pre}.lm_head.{j}.weight"]).data_ptr() for j in range(n)])
        t.predictor_embeddings = self._pe
        t.lm_heads = self._lh
        tc, ts = rope_tables(cfg.talker.head_dim, cfg.talker.rope_theta, self.max_seq_len, self.dtype)
        pc, ps = rope_tables(cfg.predictor.head_dim, cfg.predictor.rope_theta, cfg.num_code_groups + 1, self.dtype)
        self._rope = [x.to(self.device) for x in (tc, ts, pc, ps)]
        t.talker_cos, t.talker_sin, t.talker_rope_len = self._rope[0].data_ptr(), self._rope[1].data_ptr(), self.max_seq_len
        t.pred_cos, t.pred_sin, t.pred_rope_len = self._rope[2].data_ptr(), self._rope[3].data_ptr(), cfg.num_code_groups + 1
        self._table = t
        L.check(self.lib.fq3_bind_weights(self.ctx, C.byref(t)))

    # ------------------------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _chk(self, t: torch.Tensor, n: int, what: str):
        if t.device != self.device and not (t.device.type == "cuda" and self.device.index in (None, t.device.index)):
            raise ValueError(f"{what}: tensor is on {t.device}, context on {self.device}")
        if t.dtype != self.dtype or not t.is_contiguous() or t.numel() != n:
            raise ValueError(f"{what}: need contiguous {self.dtype} tensor with {n} elements, got {t.dtype} {tuple(t.shape)}")

    def new(self, *shape, dtype=None) -> torch.Tensor:
        return torch.empty(*shape, dtype=dtype or self.dtype, device=self.device)

    # ---- talker ----
    def set_generation_state(self, n_pad: int, rope_delta: int):
        L.check(self.lib.fq3_set_generation_state(self.ctx, int(n_pad), int(rope_delta)))

    def kv_import(self, layer: int, k: torch.Tensor, v: torch.Tensor):
        Lk = k.shape[-2]
        k = k.to(self.dtype).contiguous(); v = v.to(self.dtype).contiguous()
        L.check(self.lib.fq3_kv_import(self.ctx, layer, k.data_ptr(), v.data_ptr(), int(Lk), self._stream()))

    def kv_export(self, layer: int, Lk: int):
        nk, d = self.cfg.talker.num_key_value_heads, self.cfg.talker.head_dim
        k, v = self.new(nk, Lk, d), self.new(nk, Lk, d)
        L.check(self.lib.fq3_kv_export(self.ctx, layer, k.data_ptr(), v.data_ptr(), int(Lk), self._stream()))
        return k, v

    def kv_adopt(self, src: "Fq3Engine", Lk: int):
        """Take over the first ``Lk`` KV rows of every talker layer from another context (``fq3_kv_adopt``): a block-table
        hand-over when both draw from one pool (``src`` is left empty), a block-wise copy otherwise."""
        L.check(self.lib.fq3_kv_adopt(self.ctx, src.ctx, int(Lk), self._stream()))

    def kv_reserve(self, n_positions: int):
        """Take the KV blocks of key slots ``[0, n_positions)`` now (``fq3_kv_reserve``); ``Fq3Error(FQ3_ENOMEM)`` if the pool is
        short (nothing is taken then)."""
        L.check(self.lib.fq3_kv_reserve(self.ctx, int(n_positions), self._stream()))

    def kv_release(self, keep_positions: int = 0):
        """Return the KV blocks beyond ``keep_positions`` key slots to the pool (``fq3_kv_release``).  The caller makes sure this
        context's queued work has finished."""
        L.check(self.lib.fq3_kv_release(self.ctx, int(keep_positions)))

    def kv_blocks(self) -> int:
        """64-key blocks this context owns now."""
        return int(self.lib.fq3_kv_blocks(self.ctx))

    def decode_cancel(self):
        """Mark the on-device loop done (``fq3_decode_cancel``): an abandoned utterance stops at the next frame boundary."""
        L.check(self.lib.fq3_decode_cancel(self.ctx, self._stream()))

    def talker_step(self, embeds: torch.Tensor, position: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        H = self.cfg.talker.hidden_size
        self._chk(embeds, H, "talker_step embeds")
        out = out if out is not None else self.new(H)
        L.check(self.lib.fq3_talker_step(self.ctx, embeds.data_ptr(), int(position), out.data_ptr(), self._stream()))
        return out

    def prefill(self, embeds: torch.Tensor, n_pad: int = 0):
        """embeds [L, H] -> (logits [V], hidden [H])."""
        Lp, H = embeds.shape
        self._chk(embeds, Lp * H, "prefill embeds")
        logits, hid = self.new(self.cfg.talker.vocab_size), self.new(H)
        L.check(self.lib.fq3_prefill(self.ctx, embeds.data_ptr(), int(Lp), int(n_pad), logits.data_ptr(),
                                     hid.data_ptr(), self._stream()))
        return logits, hid

    def prefill_reserve(self):
        """Allocate this context's prefill workspace now (``fq3_prefill_reserve``) instead of inside its first prefill."""
        with torch.cuda.device(self.device):
            L.check(self.lib.fq3_prefill_reserve(self.ctx))

    @staticmethod
    def prefill_batch(engines, embeds, n_pads=None):
        """Several prompts in one pass over the weights (``fq3_prefill_batch``): ``embeds[i]`` [L_i, H] is prefilled into
        ``engines[i]``'s cache.  Returns ``[(logits [V], hidden [H]), ...]``.  The engines must share one weight table."""
        import ctypes as C
        e0 = engines[0]
        n = len(engines)
        n_pads = list(n_pads) if n_pads is not None else [0] * n
        H, V = e0.cfg.talker.hidden_size, e0.cfg.talker.vocab_size
        outs = []
        for e, x in zip(engines, embeds):
            e._chk(x, x.shape[0] * H, "prefill embeds")
            outs.append((e.new(V), e.new(H)))
        vp = C.c_void_p
        ctxs = (vp * n)(*[e.ctx for e in engines])
        emb = (vp * n)(*[x.data_ptr() for x in embeds])
        Ls = (C.c_int * n)(*[int(x.shape[0]) for x in embeds])
        pads = (C.c_int * n)(*[int(p) for p in n_pads])
        lg = (vp * n)(*[o[0].data_ptr() for o in outs])
        hd = (vp * n)(*[o[1].data_ptr() for o in outs])
        L.check(e0.lib.fq3_prefill_batch(ctxs, n, emb, Ls, pads, lg, hd, e0._stream()))
        return outs

    def set_option(self, key: str, value: int):
        """Kernel-variant switch (``fq3_set_option``): weight_nt, pred_m2, pred_attn, rows_per_wave_max, prefill_mode, flash_prefill,
        flash_small, skinny_gemm."""
        L.check(self.lib.fq3_set_option(self.ctx, key.encode(), int(value)))

    def decode_set_forced(self, forced: Optional[torch.Tensor], decisions: Optional[torch.Tensor]):
        """Teacher-forcing test hook (``fq3_decode_set_forced``): int32 device tensors [frames + 1, 16] or None."""
        for t in (forced, decisions):
            if t is not None and (t.dtype != torch.int32 or not t.is_contiguous() or not t.is_cuda):
                raise ValueError("forced / decisions must be contiguous int32 device tensors")
        self._forced_keep = (forced, decisions)
        L.check(self.lib.fq3_decode_set_forced(self.ctx, _ptr(forced), _ptr(decisions), self._stream()))

    def set_prefill_mode(self, mode: int):
        """0 = MFMA prefill (default), 1 = token-by-token walk through the decode kernels (test hook)."""
        L.check(self.lib.fq3_set_prefill_mode(self.ctx, int(mode)))

    def codec_head(self, hidden: torch.Tensor) -> torch.Tensor:
        self._chk(hidden, self.cfg.talker.hidden_size, "codec_head hidden")
        out = self.new(self.cfg.talker.vocab_size)
        L.check(self.lib.fq3_codec_head(self.ctx, hidden.data_ptr(), out.data_ptr(), self._stream()))
        return out

    # ---- prompt builder ----
    def bind_prompt_weights(self, text_embedding: torch.Tensor, fc1_w: torch.Tensor, fc1_b: torch.Tensor,
                            fc2_w: torch.Tensor, fc2_b: torch.Tensor):
        """talker.get_text_embeddings() / talker.text_projection weights for ``text_project`` (tensors are borrowed)."""
        ts = [t.to(device=self.device, dtype=self.dtype).contiguous() for t in (text_embedding, fc1_w, fc1_b, fc2_w, fc2_b)]
        self._prompt_keep = ts
        w = L.PromptWeights(*[t.data_ptr() for t in ts], int(ts[0].shape[0]), int(ts[0].shape[1]))
        L.check(self.lib.fq3_bind_prompt_weights(self.ctx, C.byref(w)))

    def text_project(self, ids: torch.Tensor) -> torch.Tensor:
        """text_projection(text_embedding(ids)) for a flat LongTensor of token ids -> [n, H]."""
        ids = ids.reshape(-1).to(device=self.device, dtype=torch.long).contiguous()
        out = self.new(ids.numel(), self.cfg.talker.hidden_size)
        L.check(self.lib.fq3_text_project(self.ctx, ids.data_ptr(), int(ids.numel()), out.data_ptr(), self._stream()))
        return out

    def prompt_rows(self, text_rows: Optional[torch.Tensor], prog: torch.Tensor, ref_codes: Optional[torch.Tensor] = None,
                    spk_embed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Assemble prompt rows from a row program int32[n, 3] = (text_row | -1, kind, arg); see ``fq3_prompt_rows``."""
        H = self.cfg.talker.hidden_size
        prog = prog.to(device=self.device, dtype=torch.int32).contiguous()
        n = prog.shape[0]
        if text_rows is not None:
            self._chk(text_rows, text_rows.shape[0] * H, "prompt_rows text_rows")
        if spk_embed is not None:
            spk_embed = spk_embed.reshape(-1).to(device=self.device, dtype=self.dtype).contiguous()
            self._chk(spk_embed, H, "prompt_rows spk_embed")
        if ref_codes is not None:
            ref_codes = ref_codes.to(device=self.device, dtype=torch.long).contiguous()
        out = self.new(n, H)
        L.check(self.lib.fq3_prompt_rows(self.ctx, _ptr(text_rows), int(text_rows.shape[0]) if text_rows is not None else 0,
                                         prog.data_ptr(), int(n), _ptr(ref_codes),
                                         int(ref_codes.shape[0]) if ref_codes is not None else 0, _ptr(spk_embed),
                                         out.data_ptr(), self._stream()))
        return out

    # ---- predictor ----
    def set_predictor_sampling(self, *, do_sample: bool, top_k: int, top_p: float, temperature: float):
        s = L.Sampling(float(temperature), int(top_k), float(top_p), int(bool(do_sample)), 1.0)
        L.check(self.lib.fq3_set_predictor_sampling(self.ctx, C.byref(s)))
        self._pred_sampling = dict(do_sample=bool(do_sample), top_k=int(top_k), top_p=float(top_p),
                                   temperature=float(temperature))

    def predictor_loop(self, pred_input: torch.Tensor, noise: Optional[torch.Tensor] = None,
                       want_logits: bool = False):
        H, n, Vp = self.cfg.talker.hidden_size, self.cfg.num_code_groups - 1, self.cfg.predictor.vocab_size
        self._chk(pred_input, 2 * H, "predictor_loop pred_input")
        if noise is not None:
            self._chk(noise, n * Vp, "predictor_loop noise")
        ids = torch.empty(n, dtype=torch.long, device=self.device)
        lg = self.new(n, Vp) if want_logits else None
        L.check(self.lib.fq3_predictor_loop(self.ctx, pred_input.data_ptr(), _ptr(noise), ids.data_ptr(), _ptr(lg),
                                            self._stream()))
        return (ids, lg) if want_logits else ids

    # ---- sampler ----
    def sample(self, logits: torch.Tensor, *, temperature: float, top_k: int, top_p: float, do_sample: bool,
               repetition_penalty: float = 1.0, history: Optional[torch.Tensor] = None, sup_lo: int = 0,
               sup_hi: int = 0, keep_id: int = -1, suppress_eos: bool = False,
               noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        V = logits.numel()
        self._chk(logits, V, "sample logits")
        if noise is not None:
            self._chk(noise, V, "sample noise")
        s = L.Sampling(float(temperature), int(top_k), float(top_p), int(bool(do_sample)), float(repetition_penalty))
        out = torch.empty(1, dtype=torch.long, device=self.device)
        nh = 0
        if history is not None and history.numel() > 0:
            history = history.to(device=self.device, dtype=torch.long).contiguous()
            nh = history.numel()
        else:
            history = None
        L.check(self.lib.fq3_sample(self.ctx, logits.data_ptr(), int(V), C.byref(s), _ptr(history), nh, int(sup_lo),
                                    int(sup_hi), int(keep_id), int(bool(suppress_eos)), _ptr(noise), out.data_ptr(),
                                    self._stream()))
        return out

    # ---- fused on-device loop ----
    def decode_begin(self, *, first_token: int, prefill_len: int, gen_step: int, past_hidden: torch.Tensor,
                     trailing_text: torch.Tensor, tts_pad_embed: torch.Tensor, temperature: float, top_k: int,
                     top_p: float, do_sample: bool, repetition_penalty: float, min_new_tokens: int,
                     max_new_tokens: int, talker_noise: Optional[torch.Tensor] = None,
                     pred_noise: Optional[torch.Tensor] = None, noise_frames: int = 0):
        H = self.cfg.talker.hidden_size
        self._chk(past_hidden, H, "past_hidden")
        self._chk(tts_pad_embed, H, "tts_pad_embed")
        tl = trailing_text.shape[-2] if trailing_text is not None and trailing_text.numel() else 0
        if tl:
            self._chk(trailing_text, tl * H, "trailing_text")
        p = L.DecodeParams()
        p.talker = L.Sampling(float(temperature), int(top_k), float(top_p), int(bool(do_sample)), float(repetition_penalty))
        p.min_new_tokens, p.max_new_tokens = int(min_new_tokens), int(max_new_tokens)
        p.prefill_len, p.gen_step, p.first_token = int(prefill_len), int(gen_step), int(first_token)
        p.past_hidden, p.trailing_text, p.trailing_len = past_hidden.data_ptr(), _ptr(trailing_text) if tl else None, tl
        p.tts_pad_embed = tts_pad_embed.data_ptr()
        p.talker_noise, p.pred_noise, p.noise_frames = _ptr(talker_noise), _ptr(pred_noise), int(noise_frames)
        self._loop_keep = (past_hidden, trailing_text, tts_pad_embed, talker_noise, pred_noise)
        L.check(self.lib.fq3_decode_begin(self.ctx, C.byref(p), self._stream()))

    def graph_capture(self):
        with _CAPTURE_LOCK:          # one capture at a time per process (several contexts may share a GPU)
            L.check(self.lib.fq3_graph_capture(self.ctx, self._stream()))

    def graph_reset(self):
        L.check(self.lib.fq3_graph_reset(self.ctx))

    def decode_frames(self, n: int):
        L.check(self.lib.fq3_decode_frames(self.ctx, int(n), self._stream()))

    def decode_poll(self):
        n, d = C.c_int(0), C.c_int(0)
        L.check(self.lib.fq3_decode_poll(self.ctx, C.byref(n), C.byref(d), self._stream()))
        return n.value, bool(d.value)

    def decode_codes(self, start: int, count: int) -> torch.Tensor:
        out = torch.empty(count, self.cfg.num_code_groups, dtype=torch.long, device=self.device)
        if count:
            L.check(self.lib.fq3_decode_codes(self.ctx, int(start), int(count), out.data_ptr(), self._stream()))
        return out

    def close(self):
        if getattr(self, "ctx", None) and self.ctx.value:
            self.lib.fq3_ctx_destroy(self.ctx)
            self.ctx = L.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Fq3Batch:
    """B decode contexts ("lanes") advanced in lock-step by ONE launch chain and ONE pass over the weights per frame
    (``fq3_batch_*``; no reference equivalent -- the reference fixes batch = 1, talker_graph.py:46).

    Every lane is an ordinary :class:`Fq3Engine` sharing the first lane's weights (``share=``): prefill, generation
    state and ``decode_begin`` happen per lane, ``frames(n)`` advances all lanes, ``decode_poll`` / ``decode_codes`` are
    read per lane.  A finished (or never begun) lane idles on device and can be re-armed with ``decode_begin`` at any
    frame boundary.  A lane's ids are bit-identical to the same utterance decoded alone with the same noise."""

    def __init__(self, lanes):
        lanes = list(lanes)
        if not 1 <= len(lanes) <= 128:
            raise ValueError("a batch holds 1..128 lanes")
        self.lanes = lanes
        self.lib = lanes[0].lib
        self.device = lanes[0].device
        arr = (L.vp * len(lanes))(*[e.ctx.value for e in lanes])
        self.handle = L.vp()
        with torch.cuda.device(self.device):
            L.check(self.lib.fq3_batch_create(arr, len(lanes), C.byref(self.handle)))

    def __len__(self):
        return len(self.lanes)

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def frames(self, n: int):
        L.check(self.lib.fq3_batch_frames(self.handle, int(n), self._stream()))

    def graph_capture(self):
        with _CAPTURE_LOCK:
            L.check(self.lib.fq3_batch_graph_capture(self.handle, self._stream()))

    def graph_reset(self):
        L.check(self.lib.fq3_batch_graph_reset(self.handle))

    def set_option(self, key: str, value: int):
        """``fq3_batch_set_option``: "mfma" 0|1 (matrix-core batch GEMVs, bf16), "skinny" 0|1 (o_proj / down of 17..32 lanes on
        the weight-stationary prefill kernel), "groups" 0..4 (lane groups advanced concurrently; 0 = automatic), "norm_skinny" 0|1 (above 32 lanes: normalise once + weight-stationary
        GEMM for qkv / gate | up / heads)."""
        L.check(self.lib.fq3_batch_set_option(self.handle, key.encode(), int(value)))

    def poll_async(self, slot: int):
        """``fq3_batch_poll_async``: queue a poll of every lane (one launch, one copy) in stream order into ``slot`` (0..3)."""
        L.check(self.lib.fq3_batch_poll_async(self.handle, int(slot), self._stream()))

    def poll_wait(self, slot: int):
        """``fq3_batch_poll_wait``: ``([frames done per lane], [finished per lane])`` of the poll queued in ``slot``."""
        n = len(self.lanes)
        a, d = (C.c_int * n)(), (C.c_int * n)()
        L.check(self.lib.fq3_batch_poll_wait(self.handle, int(slot), a, d))
        return list(a), [bool(x) for x in d]

    def set_group_streams(self, streams):
        """``fq3_batch_set_group_streams``: the caller's own side streams for lane groups 1.. (kept alive here)."""
        self._group_streams = list(streams)
        arr = (L.vp * max(1, len(self._group_streams)))(*[s.cuda_stream for s in self._group_streams])
        L.check(self.lib.fq3_batch_set_group_streams(self.handle, arr, len(self._group_streams)))

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.fq3_batch_destroy(self.handle)
            self.handle = L.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

