"""HIP 12 Hz codec decoder behind the upstream ``speech_tokenizer`` duck type.

The reference calls ``speech_tokenizer.decode({"audio_codes": LongTensor[B, T, 16]}) ->
(list[1-D float waveform], sample_rate)`` (``faster_qwen3_tts/model.py:924``; payload shape pinned by the
reference's ``tests/test_sample_rate.py:53-75``) and reads ``.sample_rate``.  This class keeps that
surface and runs the whole decoder in ``libfq3hip.so``; weights are re-laid-out once at load time
into the GEMM-ready layouts documented in ``csrc/fq3_codec.hip``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import torch

from . import _lib as L
from .config import CodecConfig
from .engine import rope_tables

Weights = Dict[str, torch.Tensor]


def pack_codec_weights(W: Weights, c: CodecConfig) -> Weights:
    """checkpoint layout -> kernel layout (host glue, runs once)."""
    out: Weights = {}
    p = "decoder"
    conv = lambda w: w.permute(0, 2, 1).contiguous()                      # [Cout, Cin, k] -> [Cout, k, Cin]

    def convT(w, s):                                                      # [Cin, Cout, k] -> [q*Cout+co, tap, Cin]
        cin, cout, k = w.shape
        taps = k // s
        w5 = w.reshape(cin, cout, taps, s)                                # k = tap*s + q
        return w5.permute(3, 1, 2, 0).reshape(s * cout, taps, cin).contiguous()

    for k, v in W.items():
        if not k.startswith(p + "."):
            continue
        out[k] = v
    out[f"{p}.pre_conv.conv.weight"] = conv(W[f"{p}.pre_conv.conv.weight"])
    for name in ("rvq_first", "rvq_rest"):
        out[f"{p}.quantizer.{name}.output_proj.weight"] = W[f"{p}.quantizer.{name}.output_proj.weight"].squeeze(-1).contiguous()
    t = f"{p}.pre_transformer"
    for i in range(c.num_hidden_layers):
        q = f"{t}.layers.{i}"
        out[f"{q}.self_attn.qkv.weight"] = torch.cat([W[f"{q}.self_attn.q_proj.weight"], W[f"{q}.self_attn.k_proj.weight"],
                                                      W[f"{q}.self_attn.v_proj.weight"]], 0).contiguous()
        # gate / up rows interleaved in groups of 16 so that one MFMA column-tile pair holds (gate, up) of the same 16
        # logical columns and SwiGLU becomes the GEMM's epilogue
        gp, up = W[f"{q}.mlp.gate_proj.weight"], W[f"{q}.mlp.up_proj.weight"]
        I, H = gp.shape
        out[f"{q}.mlp.gate_up.weight"] = torch.stack([gp.reshape(I // 16, 16, H), up.reshape(I // 16, 16, H)], 1).reshape(2 * I, H).contiguous()
    for i, f in enumerate(c.upsampling_ratios):
        u = f"{p}.upsample.{i}"
        out[f"{u}.0.conv.weight"] = convT(W[f"{u}.0.conv.weight"], f)
        out[f"{u}.1.dwconv.conv.weight"] = W[f"{u}.1.dwconv.conv.weight"].reshape(-1, 7).contiguous()
    d = f"{p}.decoder"
    out[f"{d}.0.conv.weight"] = conv(W[f"{d}.0.conv.weight"])
    for i, r in enumerate(c.upsample_rates):
        b = f"{d}.{i + 1}.block"
        out[f"{b}.1.conv.weight"] = convT(W[f"{b}.1.conv.weight"], r)
        for j in (2, 3, 4):
            out[f"{b}.{j}.conv1.conv.weight"] = conv(W[f"{b}.{j}.conv1.conv.weight"])
            out[f"{b}.{j}.conv2.conv.weight"] = conv(W[f"{b}.{j}.conv2.conv.weight"])
    n = len(c.upsample_rates)
    out[f"{d}.{n + 2}.conv.weight"] = W[f"{d}.{n + 2}.conv.weight"][0].transpose(0, 1).contiguous()   # [C,7] -> [7,C]
    return out


_GEMM_SUFFIXES = ("output_proj.weight", "input_proj.weight", "qkv.weight", "o_proj.weight", "gate_up.weight", "down_proj.weight",
                  "pwconv1.weight", "pwconv2.weight")
_UNPACKED = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "gate_proj.weight", "up_proj.weight")      # superseded by qkv / gate_up


def _is_gemm_weight(name: str, c: CodecConfig) -> bool:
    """Is this packed tensor the B operand of a matrix-core GEMM (as opposed to a per-channel vector / table an elementwise kernel
    reads)?  Follows the binding contract documented in csrc/fq3_codec.hip."""
    if name.endswith(_GEMM_SUFFIXES):
        return True
    final = f"decoder.decoder.{len(c.upsample_rates) + 2}.conv.weight"
    return name.endswith(".conv.weight") and ".dwconv." not in name and name != final


def to_bf16x2(x: torch.Tensor) -> torch.Tensor:
    """fp32 values -> int32 words ``bf16(x) | bf16(x - bf16(x)) << 16``: the storage type of the codec's high-precision mode
    (``bfs_t`` of csrc/fq3_common.cuh; 16 mantissa bits)."""
    x = x.float().contiguous()
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    hi_u = hi.view(torch.int16).to(torch.int32) & 0xFFFF
    lo_u = lo.view(torch.int16).to(torch.int32) & 0xFFFF
    return (hi_u | (lo_u << 16)).contiguous()


class HipSpeechTokenizer:
    """``speech_tokenizer`` replacement whose ``decode`` runs on the HIP codec kernels.

    ``precision`` (default: follows ``dtype``): ``"bf16"`` -- the checkpoint dtype, what the reference's Torch path runs; ``"fp32"`` --
    weights widened exactly, fp32 activations and fp32 matrix-core products; ``"bf16x2"`` -- the high-precision mode for serving:
    bf16 weights, every activation kept as a bf16 high part + a bf16 residual, two bf16 MFMAs per product pair (``FQ3_BF16X2``):
    within the north star's 1e-3 PCM RMS of an fp32 evaluation at about twice the bf16 decode time."""

    def __init__(self, cfg: CodecConfig, weights: Weights, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                 max_frames: int = 1024, share: "HipSpeechTokenizer" = None, precision: str = None):
        self.lib = L.load()
        if precision is None:
            precision = "bf16" if dtype == torch.bfloat16 else "fp32"
        if precision not in ("bf16", "fp32", "bf16x2"):
            raise ValueError("codec precision must be 'bf16', 'fp32' or 'bf16x2'")
        dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "bf16x2": torch.float32}[precision]
        self.precision = precision
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.sample_rate = int(cfg.sample_rate)
        self.max_frames = int(max_frames)
        cc = L.CodecConfig()
        cc.dtype = {"bf16": L.FQ3_BF16, "fp32": L.FQ3_F32, "bf16x2": L.FQ3_BF16X2}[precision]
        cc.codebook_size, cc.codebook_dim, cc.rvq_dim = cfg.codebook_size, cfg.codebook_dim, cfg.rvq_dim
        cc.num_quantizers, cc.num_semantic = cfg.num_quantizers, cfg.num_semantic_quantizers
        cc.latent_dim, cc.hidden, cc.inter = cfg.latent_dim, cfg.hidden_size, cfg.intermediate_size
        cc.n_layers, cc.n_heads, cc.head_dim = cfg.num_hidden_layers, cfg.num_attention_heads, cfg.head_dim
        cc.sliding_window, cc.rms_eps = cfg.sliding_window, cfg.rms_norm_eps
        cc.n_upsample = len(cfg.upsampling_ratios)
        for i, v in enumerate(cfg.upsampling_ratios):
            cc.upsampling_ratios[i] = v
        cc.n_rates = len(cfg.upsample_rates)
        for i, v in enumerate(cfg.upsample_rates):
            cc.upsample_rates[i] = v
        cc.decoder_dim, cc.max_frames = cfg.decoder_dim, self.max_frames
        self.h = L.vp()
        with torch.cuda.device(self.device):
            L.check(self.lib.fq3_codec_create(C.byref(cc), C.byref(self.h)))
        self._bound: Dict[str, torch.Tensor] = {}
        self._numel: Dict[str, int] = {}                # logical element counts (what fq3_codec_bind checks)
        if (share is not None and getattr(share, "precision", None) == precision and share.device == self.device
                and share.max_frames == self.max_frames):
            self._bound, self._numel = dict(share._bound), dict(share._numel)       # borrow the packed tensors of another decoder instance
        else:
            for name, t in pack_codec_weights(weights, cfg).items():
                self._numel[name] = t.numel()
                if precision != "bf16x2":
                    self._bound[name] = t.to(device=self.device, dtype=dtype).contiguous()
                elif name.endswith(_UNPACKED):
                    del self._numel[name]               # only their packed forms are read
                elif _is_gemm_weight(name, cfg):
                    # B operand: bf16, every K column twice -- the [rows][2 Cin] bf16 image of a bf16x2 activation holds (hi, lo) pairs
                    w = t.to(device=self.device, dtype=torch.bfloat16)
                    self._bound[name] = torch.repeat_interleave(w, 2, dim=-1).contiguous()
                else:
                    self._bound[name] = to_bf16x2(t.to(self.device))
            cs, sn = rope_tables(cfg.head_dim, cfg.rope_theta, self.max_frames, dtype)
            self._bound["rope.cos"] = cs.to(self.device).contiguous()
            self._bound["rope.sin"] = sn.to(self.device).contiguous()
            self._numel["rope.cos"], self._numel["rope.sin"] = cs.numel(), sn.numel()
        for name, t in self._bound.items():
            L.check(self.lib.fq3_codec_bind(self.h, name.encode(), t.data_ptr(), self._numel[name]))
        with torch.cuda.device(self.device):
            L.check(self.lib.fq3_codec_finalize(self.h, torch.cuda.current_stream(self.device).cuda_stream))

    def attach_encoder(self, analyzer) -> None:
        """Give the tokenizer its encode half (``fq3hip.refenc.HipRefAudioAnalyzer``)."""
        self._analyzer = analyzer

    def encode(self, audio, sr: int = None):
        """Upstream ``speech_tokenizer.encode(audio, sr)`` [recalled]: -> object with ``audio_codes`` = list of
        ``LongTensor[T, 16]``.  ``audio`` is one waveform (or a list of them) at ``sr`` Hz."""
        from types import SimpleNamespace
        an = getattr(self, "_analyzer", None)
        if an is None:
            raise NotImplementedError("this speech tokenizer was built without encoder.* weights")
        from .audio_io import resample
        import numpy as np
        items = audio if isinstance(audio, (list, tuple)) else [audio]
        out = []
        for a in items:
            a = np.asarray(a.detach().cpu().numpy() if hasattr(a, "detach") else a, dtype=np.float32).reshape(-1)
            out.append(an.encode(resample(a, int(sr or self.sample_rate), self.sample_rate)))
        return SimpleNamespace(audio_codes=out)

    def set_option(self, key: str, value: int):
        """``fq3_codec_set_option``: "fuse_units" 0|1|2 (two GEMMs per residual unit | the 96-channel block's units fused into one launch
        each, the default | the 192-channel block's too); "glds_cap8" n (measurement hook: bf16x2 GEMMs on the eight-wave LDS-DMA tiles up
        to n tiles; 0 = always); all settings give bit-identical waveforms."""
        L.check(self.lib.fq3_codec_set_option(self.h, key.encode(), int(value)))

    def num_samples(self, n_frames: int) -> int:
        return int(self.lib.fq3_codec_num_samples(self.h, int(n_frames)))

    # upstream decodes long inputs in pieces (transformers sibling ``chunked_decode``,
    # modeling_qwen3_omni_moe.py:3686-3696: chunk_size=300, left_context_size=25)
    CHUNK_FRAMES = 300
    LEFT_CONTEXT = 25

    # ---- reference-prefix states (``fq3_codec_prefix_*``, round 6) -------------------------------------------------------------
    # The ICL call sites decode ``ref_codes + generated codes`` (model.py:919-937; every phase-1 streaming chunk again, :1085-1115).
    # ``prefix_for(ref_codes)`` runs the frame-level front end over the reference frames ONCE per voice and keeps what later rows
    # read of them; ``decode_tensor(..., prefix=...)`` / ``decode_tensor_batch(..., prefixes=...)`` then compute the front end over
    # the new rows only.  Bit-identical to the full decode (tests/test_gpu_codec.py).  ``use_prefix = False`` turns it off.
    use_prefix = True
    PREFIX_CACHE = 64           # voices kept (least recently used first out)

    class Prefix:
        """Owner of one ``fq3_codec_prefix`` handle (destroyed with the object or by ``HipSpeechTokenizer.close``)."""

        def __init__(self, tok, handle, ref_len: int, keep):
            self.tok, self.h, self.ref_len, self._keep = tok, handle, int(ref_len), keep

        def close(self):
            if getattr(self, "h", None) is not None and self.h.value:
                self.tok.lib.fq3_codec_prefix_destroy(self.h)
                self.h = L.vp()

        def __del__(self):
            try:
                self.close()
            except Exception:
                pass

    def prefix_for(self, ref_codes, stream=None):
        """The (cached) prefix state of ``ref_codes`` LongTensor[ref_len, 16], or None (no reference / switched off / not applicable).
        A device tensor is recognised by identity (address, shape, version counter -- the entry keeps it alive), a host tensor by
        its content.  The state is built on ``stream`` (a ``torch.cuda.Stream``; default: the current one): use the stream the
        decodes run on -- the codec's workspaces are shared by everything that goes through this tokenizer."""
        if not self.use_prefix or ref_codes is None or not hasattr(ref_codes, "shape") or ref_codes.dim() != 2:
            return None
        n = int(ref_codes.shape[0])
        if n < 1 or n >= min(self.CHUNK_FRAMES, self.max_frames):
            return None
        if ref_codes.device.type == "cpu":
            import hashlib
            key = ("host", n, hashlib.blake2b(ref_codes.contiguous().numpy().tobytes(), digest_size=16).digest())
        else:
            # (a tensor made under torch.inference_mode() has no version counter: -1)
            key = ("dev", int(ref_codes.data_ptr()), tuple(ref_codes.shape), str(ref_codes.dtype),
                   -1 if ref_codes.is_inference() else int(ref_codes._version))
        cache = self.__dict__.setdefault("_prefixes", {})
        hit = cache.pop(key, None)
        if hit is not None:
            cache[key] = hit                                       # most recently used last
            return hit
        codes = ref_codes.to(device=self.device, dtype=torch.long).contiguous()
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream(self.device))
            codes.record_stream(stream)
        h = L.vp()
        with torch.cuda.stream(st):
            L.check(self.lib.fq3_codec_prefix_create(self.h, codes.data_ptr(), n, C.byref(h), st.cuda_stream))
        pf = HipSpeechTokenizer.Prefix(self, h, n, (ref_codes, codes))
        cache[key] = pf
        while len(cache) > self.PREFIX_CACHE:
            cache.pop(next(iter(cache)))                           # (dropped states die with their last user)
        return pf

    def _decode_piece(self, codes: torch.Tensor, first_sample: int = 0, prefix=None) -> torch.Tensor:
        """One decoder pass over codes [T <= max_frames, 16]; returns samples [first_sample, num_samples(T))."""
        if prefix is not None and prefix.ref_len < codes.shape[0]:
            return self._decode_piece_batch(codes.unsqueeze(0), first_sample, [prefix])[0]
        Tn = codes.shape[0]
        n = self.num_samples(Tn)
        first_sample = max(0, min(int(first_sample), n))
        pcm = torch.empty(n - first_sample, dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        if first_sample == 0:
            L.check(self.lib.fq3_codec_decode(self.h, codes.data_ptr(), int(Tn), pcm.data_ptr(), st))
        elif first_sample < n:
            L.check(self.lib.fq3_codec_decode_tail(self.h, codes.data_ptr(), int(Tn), first_sample, pcm.data_ptr(), st))
        return pcm

    def _pieces(self, Tn: int):
        """(start, end, ctx) of every decoder pass for Tn frames: one pass up to the chunk size, otherwise upstream's
        ``chunked_decode`` schedule (CHUNK_FRAMES new frames + LEFT_CONTEXT context frames per pass)."""
        if Tn <= min(self.CHUNK_FRAMES, self.max_frames):
            return [(0, Tn, 0)]
        out, start = [], 0
        while start < Tn:
            ctx = self.LEFT_CONTEXT if start - self.LEFT_CONTEXT > 0 else start
            ctx = min(ctx, self.max_frames - 1)
            # upstream: CHUNK_FRAMES new frames per piece; a workspace smaller than 325 frames shortens the pieces
            end = min(start + min(self.CHUNK_FRAMES, self.max_frames - ctx), Tn)
            out.append((start, end, ctx))
            start = end
        return out

    def decode_tensor(self, codes: torch.Tensor, first_sample: int = 0, prefix=None) -> torch.Tensor:
        """codes LongTensor[T, 16] -> float32 waveform tensor on the device.  Any T: inputs longer than the chunk size
        are decoded piecewise with a 25-frame left context, as upstream ``chunked_decode`` does.

        ``first_sample`` > 0 returns ``decode_tensor(codes)[first_sample:]`` (bit-identical) while recomputing only the
        rows those samples depend on -- what the streaming call sites keep of each re-decode.

        ``prefix`` (``prefix_for(ref_codes)``): ``codes`` starts with that reference; the front end then runs over the rows behind it only
        (the piece that starts at frame 0 -- later pieces of a long input begin behind the reference anyway)."""
        codes = codes.to(device=self.device, dtype=torch.long).contiguous()
        Tn = codes.shape[0]
        up = self.cfg.total_upsample
        wavs, pos = [], 0                      # pos = index (in the concatenated waveform) of the piece's first kept sample
        for start, end, ctx in self._pieces(Tn):
            n_piece = self.num_samples(end - start + ctx) - ctx * up
            if pos + n_piece > first_sample:
                skip = ctx * up + max(0, first_sample - pos)
                wavs.append(self._decode_piece(codes[start - ctx:end].contiguous(), skip, prefix if start == 0 else None))
            pos += n_piece
        if not wavs:
            return torch.empty(0, dtype=torch.float32, device=self.device)
        return torch.cat(wavs) if len(wavs) != 1 else wavs[0]

    def _decode_piece_batch(self, codes: torch.Tensor, first_sample: int = 0, prefixes=None) -> torch.Tensor:
        """One decoder pass over codes [B, T <= max_frames, 16] (``fq3_codec_decode_batch``): [B, num_samples(T) - first_sample].
        ``prefixes``: B prefix states of one length that the utterances start with (``fq3_codec_decode_batch_prefix``), or None."""
        B, Tn = codes.shape[0], codes.shape[1]
        n = self.num_samples(Tn)
        first_sample = max(0, min(int(first_sample), n))
        pcm = torch.empty(B, n - first_sample, dtype=torch.float32, device=self.device)
        if prefixes is not None and (len(prefixes) != B or B > 128 or any(p is None or p.ref_len != prefixes[0].ref_len for p in prefixes)
                                     or prefixes[0].ref_len >= Tn):
            prefixes = None
        if first_sample < n and prefixes is not None:
            codes = codes.contiguous()
            arr = (L.vp * B)(*[p.h for p in prefixes])
            L.check(self.lib.fq3_codec_decode_batch_prefix(self.h, arr, codes.data_ptr(), int(B), int(Tn), int(first_sample), pcm.data_ptr(),
                                                           torch.cuda.current_stream(self.device).cuda_stream))
            return pcm
        if first_sample < n:
            rc = self.lib.fq3_codec_decode_batch(self.h, codes.data_ptr(), int(B), int(Tn), int(first_sample), pcm.data_ptr(),
                                                 torch.cuda.current_stream(self.device).cuda_stream)
            if rc == L.FQ3_ENOMEM and B > 1:
                # the workspace for B utterances of this length could not be grown: the same waveforms one utterance at a time (a row of the
                # batched decode IS the single decode, bit for bit), in the workspace that exists
                for b in range(B):
                    L.check(self.lib.fq3_codec_decode_tail(self.h, codes[b].data_ptr(), int(Tn), int(first_sample), pcm[b].data_ptr(),
                                                           torch.cuda.current_stream(self.device).cuda_stream))
                return pcm
            L.check(rc)
        return pcm

    def decode_tensor_batch(self, codes: torch.Tensor, first_sample: int = 0, prefixes=None) -> torch.Tensor:
        """codes LongTensor[B, T, 16] -> float32 [B, samples] on the device: row b is, bit for bit, ``decode_tensor(codes[b],
        first_sample)``, but the B utterances share every launch (one ``[B, T, 16]`` vocoder call as the reference's interface takes it,
        model.py:924).  Utterances of different lengths: pad to the longest with valid ids and keep ``num_samples_total(T_b)`` samples
        of each -- the decoder is causal."""
        codes = codes.to(device=self.device, dtype=torch.long).contiguous()
        if codes.dim() != 3:
            raise ValueError("codes must be [B, T, num_quantizers]")
        B, Tn = codes.shape[0], codes.shape[1]
        up = self.cfg.total_upsample
        wavs, pos = [], 0
        for start, end, ctx in self._pieces(Tn):
            n_piece = self.num_samples(end - start + ctx) - ctx * up
            if pos + n_piece > first_sample:
                skip = ctx * up + max(0, first_sample - pos)
                wavs.append(self._decode_piece_batch(codes[:, start - ctx:end].contiguous(), skip, prefixes if start == 0 else None))
            pos += n_piece
        if not wavs:
            return torch.empty(B, 0, dtype=torch.float32, device=self.device)
        return torch.cat(wavs, dim=1) if len(wavs) != 1 else wavs[0]

    def num_samples_total(self, Tn: int) -> int:
        """Length of ``decode_tensor(codes[Tn])`` (piecewise decodes drop the context samples of every later piece)."""
        up = self.cfg.total_upsample
        return sum(self.num_samples(e - s + c) - c * up for s, e, c in self._pieces(Tn))

    def decode(self, payload):
        codes = payload["audio_codes"]
        if codes.dim() != 3:
            raise ValueError("audio_codes must be [B, T, num_quantizers]")
        if codes.shape[0] == 1:
            return [self.decode_tensor(codes[0])], self.sample_rate
        wav = self.decode_tensor_batch(codes)                  # one launch set for the whole [B, T, 16] payload
        return [wav[b] for b in range(wav.shape[0])], self.sample_rate

    def close(self):
        for pf in list(self.__dict__.get("_prefixes", {}).values()):
            pf.close()                                            # a prefix state must not outlive its codec
        self.__dict__["_prefixes"] = {}
        if getattr(self, "h", None) and self.h.value:
            self.lib.fq3_codec_destroy(self.h)
            self.h = L.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
