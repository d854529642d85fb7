"""ctypes binding of ``libfq3hip.so`` (C ABI in ``include/fq3hip.h``).

The library is the product: if it is missing or fails to load, importing callers get an
``ImportError`` with build instructions -- there is deliberately no PyTorch / CPU fallback.
``import torch`` must come first so that the HIP runtime the extension resolves
(``libamdhip64.so.7``) is the very one PyTorch already loaded into the process.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads libamdhip64 first; see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libfq3hip.so")

FQ3_BF16, FQ3_F32, FQ3_BF16X2 = 0, 1, 2
FQ3_OK, FQ3_EINVAL, FQ3_EHIP, FQ3_ESTATE, FQ3_ETOOLONG, FQ3_EUNSUPPORTED, FQ3_ENOMEM = 0, -1, -2, -3, -4, -5, -6

vp = C.c_void_p
i32 = C.c_int32


class StackDims(C.Structure):
    _fields_ = [("hidden", i32), ("inter", i32), ("n_layers", i32), ("n_heads", i32), ("n_kv_heads", i32),
                ("head_dim", i32), ("vocab", i32), ("rms_eps", C.c_float)]


class Config(C.Structure):
    _fields_ = [("dtype", i32), ("talker", StackDims), ("predictor", StackDims), ("num_code_groups", i32),
                ("max_seq_len", i32), ("codec_eos_token_id", i32), ("has_projection", i32), ("max_frames", i32)]


class LayerWeights(C.Structure):
    _fields_ = [("input_norm", vp), ("qkv", vp), ("q_norm", vp), ("k_norm", vp), ("o", vp), ("post_norm", vp),
                ("gate_up", vp), ("down", vp)]


class WeightTable(C.Structure):
    _fields_ = [("talker_layers", C.POINTER(LayerWeights)), ("talker_final_norm", vp), ("codec_embedding", vp),
                ("codec_head", vp), ("predictor_layers", C.POINTER(LayerWeights)), ("predictor_final_norm", vp),
                ("proj_w", vp), ("proj_b", vp), ("predictor_embeddings", C.POINTER(vp)), ("lm_heads", C.POINTER(vp)),
                ("talker_cos", vp), ("talker_sin", vp), ("talker_rope_len", i32),
                ("pred_cos", vp), ("pred_sin", vp), ("pred_rope_len", i32)]


class Sampling(C.Structure):
    _fields_ = [("temperature", C.c_float), ("top_k", i32), ("top_p", C.c_float), ("do_sample", i32),
                ("repetition_penalty", C.c_float)]


class DecodeParams(C.Structure):
    _fields_ = [("talker", Sampling), ("min_new_tokens", i32), ("max_new_tokens", i32), ("prefill_len", i32),
                ("gen_step", i32), ("first_token", i32), ("past_hidden", vp), ("trailing_text", vp),
                ("trailing_len", i32), ("tts_pad_embed", vp), ("talker_noise", vp), ("pred_noise", vp),
                ("noise_frames", i32)]


class PromptWeights(C.Structure):
    _fields_ = [("text_embedding", vp), ("fc1_w", vp), ("fc1_b", vp), ("fc2_w", vp), ("fc2_b", vp), ("text_vocab", i32),
                ("text_hidden", i32)]


class CodecConfig(C.Structure):
    _fields_ = [("dtype", i32), ("codebook_size", i32), ("codebook_dim", i32), ("rvq_dim", i32),
                ("num_quantizers", i32), ("num_semantic", i32), ("latent_dim", i32), ("hidden", i32), ("inter", i32),
                ("n_layers", i32), ("n_heads", i32), ("head_dim", i32), ("sliding_window", i32),
                ("rms_eps", C.c_float), ("n_upsample", i32), ("upsampling_ratios", i32 * 4), ("n_rates", i32),
                ("upsample_rates", i32 * 8), ("decoder_dim", i32), ("max_frames", i32)]


class RefEncConfig(C.Structure):
    _fields_ = [("num_filters", i32), ("n_ratios", i32), ("ratios", i32 * 8), ("kernel_size", i32), ("last_kernel_size", i32),
                ("residual_kernel_size", i32), ("n_residual_layers", i32), ("dilation_growth_rate", i32), ("compress", i32),
                ("hidden", i32), ("n_layers", i32), ("n_heads", i32), ("head_dim", i32), ("inter", i32), ("sliding_window", i32),
                ("norm_eps", C.c_float), ("num_quantizers", i32), ("num_semantic", i32), ("codebook_size", i32),
                ("codebook_dim", i32), ("max_positions", i32), ("mel_dim", i32), ("n_fft", i32), ("hop", i32),
                ("n_bins_padded", i32), ("n_enc", i32), ("enc_channels", i32 * 8), ("enc_kernel_sizes", i32 * 8),
                ("enc_dilations", i32 * 8), ("attn_channels", i32), ("res2net_scale", i32), ("se_channels", i32),
                ("enc_dim", i32)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "fq3_last_error": (C.c_char_p, []),
    "fq3_abi_version": (C.c_int, []),
    "fq3_ctx_create": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
    "fq3_ctx_destroy": (C.c_int, [vp]),
    "fq3_kv_pool_create": (C.c_int, [C.POINTER(Config), C.c_int, C.POINTER(vp)]),
    "fq3_kv_pool_destroy": (C.c_int, [vp]),
    "fq3_kv_pool_stats": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "fq3_ctx_create_pooled": (C.c_int, [C.POINTER(Config), vp, C.POINTER(vp)]),
    "fq3_kv_reserve": (C.c_int, [vp, C.c_int, vp]),
    "fq3_kv_release": (C.c_int, [vp, C.c_int]),
    "fq3_kv_blocks": (C.c_int, [vp]),
    "fq3_bind_weights": (C.c_int, [vp, C.POINTER(WeightTable)]),
    "fq3_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "fq3_kv_import": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, vp]),
    "fq3_kv_export": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, vp]),
    "fq3_kv_adopt": (C.c_int, [vp, vp, C.c_int, vp]),
    "fq3_set_generation_state": (C.c_int, [vp, C.c_int, C.c_int]),
    "fq3_talker_step": (C.c_int, [vp, vp, C.c_int, vp, vp]),
    "fq3_prefill": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "fq3_prefill_reserve": (C.c_int, [vp]),
    "fq3_prefill_batch": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(vp),
                                    C.POINTER(vp), vp]),
    "fq3_set_prefill_mode": (C.c_int, [vp, C.c_int]),
    "fq3_codec_head": (C.c_int, [vp, vp, vp, vp]),
    "fq3_set_predictor_sampling": (C.c_int, [vp, C.POINTER(Sampling)]),
    "fq3_predictor_loop": (C.c_int, [vp, vp, vp, vp, vp, vp]),
    "fq3_sample": (C.c_int, [vp, vp, C.c_int, C.POINTER(Sampling), vp, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_int, vp, vp, vp]),
    "fq3_apply_repetition_penalty": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_float, vp]),
    "fq3_decode_begin": (C.c_int, [vp, C.POINTER(DecodeParams), vp]),
    "fq3_decode_cancel": (C.c_int, [vp, vp]),
    "fq3_decode_set_forced": (C.c_int, [vp, vp, vp, vp]),
    "fq3_decode_frames": (C.c_int, [vp, C.c_int, vp]),
    "fq3_decode_poll": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), vp]),
    "fq3_decode_codes": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "fq3_graph_capture": (C.c_int, [vp, vp]),
    "fq3_graph_reset": (C.c_int, [vp]),
    "fq3_bind_prompt_weights": (C.c_int, [vp, C.POINTER(PromptWeights)]),
    "fq3_text_project": (C.c_int, [vp, vp, C.c_int, vp, vp]),
    "fq3_prompt_rows": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, vp]),
    "fq3_batch_create": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(vp)]),
    "fq3_batch_destroy": (C.c_int, [vp]),
    "fq3_batch_size": (C.c_int, [vp]),
    "fq3_batch_frames": (C.c_int, [vp, C.c_int, vp]),
    "fq3_batch_graph_capture": (C.c_int, [vp, vp]),
    "fq3_batch_graph_reset": (C.c_int, [vp]),
    "fq3_batch_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "fq3_batch_set_group_streams": (C.c_int, [vp, C.POINTER(vp), C.c_int]),
    "fq3_batch_poll_async": (C.c_int, [vp, C.c_int, vp]),
    "fq3_batch_poll_wait": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fq3_codec_create": (C.c_int, [C.POINTER(CodecConfig), C.POINTER(vp)]),
    "fq3_codec_destroy": (C.c_int, [vp]),
    "fq3_codec_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "fq3_codec_bind": (C.c_int, [vp, C.c_char_p, vp, C.c_int64]),
    "fq3_codec_finalize": (C.c_int, [vp, vp]),
    "fq3_codec_num_samples": (C.c_int64, [vp, C.c_int]),
    "fq3_codec_decode": (C.c_int, [vp, vp, C.c_int, vp, vp]),
    "fq3_codec_decode_tail": (C.c_int, [vp, vp, C.c_int, C.c_int64, vp, vp]),
    "fq3_codec_decode_batch": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int64, vp, vp]),
    "fq3_codec_prefix_create": (C.c_int, [vp, vp, C.c_int, C.POINTER(vp), vp]),
    "fq3_codec_prefix_destroy": (C.c_int, [vp]),
    "fq3_codec_prefix_frames": (C.c_int, [vp]),
    "fq3_codec_decode_batch_prefix": (C.c_int, [vp, C.POINTER(vp), vp, C.c_int, C.c_int, C.c_int64, vp, vp]),
    "fq3_refenc_create": (C.c_int, [C.POINTER(RefEncConfig), C.POINTER(vp)]),
    "fq3_refenc_destroy": (C.c_int, [vp]),
    "fq3_refenc_bind": (C.c_int, [vp, C.c_char_p, vp, C.c_int64]),
    "fq3_refenc_finalize": (C.c_int, [vp, vp]),
    "fq3_refenc_num_frames": (C.c_int64, [vp, C.c_int64]),
    "fq3_refenc_encode": (C.c_int, [vp, vp, C.c_int64, vp, vp]),
    "fq3_refenc_speaker": (C.c_int, [vp, vp, C.c_int64, vp, vp, vp]),
}

_lib = None


class Fq3Error(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libfq3hip error {code}: {msg}")
        self.code = code


def load() -> C.CDLL:
    """Load the HIP library or raise ImportError (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found. Build the HIP extension first: "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C faster-qwen3-tts_amd`. "
            "There is no CPU / PyTorch fallback for the decode path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI drift; let it surface
        fn.restype = res
        fn.argtypes = args
    if lib.fq3_abi_version() != 5:
        raise ImportError("libfq3hip ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().fq3_last_error().decode("utf-8", "replace")
        if rc == FQ3_ETOOLONG:
            raise RuntimeError(msg)          # reference raises RuntimeError (talker_graph.py:163-167)
        raise Fq3Error(rc, msg)
