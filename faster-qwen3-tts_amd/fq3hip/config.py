"""Model dimensions for the Qwen3-TTS fast decode path.

Every number here is a *parameter* read from a checkpoint's ``config.json`` when one
is available (``from_hf_config``); the defaults are the recalled Qwen3-TTS-12Hz
shapes listed in SURVEY.md section 2a / Appendix D and are used only for the seeded
synthetic-weight models that the tests and ``bench.py`` run on.

Reference call sites that consume these values:
``faster_qwen3_tts/generate.py:41-43`` (eos id, code groups, vocab),
``faster_qwen3_tts/predictor_graph.py:42-46`` (predictor layers / hidden / groups),
``faster_qwen3_tts/talker_graph.py:36-37`` (talker hidden / layers).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Dict, Optional, Tuple


@dataclass
class StackConfig:
    """One pre-norm GQA transformer stack (talker backbone or code predictor)."""

    hidden_size: int = 1024
    intermediate_size: int = 3072
    num_hidden_layers: int = 28
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    vocab_size: int = 3072
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1.0e6

    @property
    def q_dim(self) -> int:
        return self.num_attention_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.num_key_value_heads * self.head_dim


@dataclass
class CodecConfig:
    """12 Hz RVQ codec decoder (vocoder).  Defaults = recalled tokenizer-v2 shapes;
    structure follows the readable sibling
    ``transformers/models/qwen3_omni_moe/modeling_qwen3_omni_moe.py:3636-3696``."""

    codebook_size: int = 2048
    codebook_dim: int = 512          # RVQ output dimension (after output_proj)
    rvq_dim: int = 256               # per-codebook embedding width (codebook_dim // 2)
    num_quantizers: int = 16
    num_semantic_quantizers: int = 1
    latent_dim: int = 1024           # conv trunk width
    hidden_size: int = 512           # transformer width
    intermediate_size: int = 1024
    num_hidden_layers: int = 8
    num_attention_heads: int = 16
    head_dim: int = 64
    sliding_window: int = 72
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    layer_scale_initial_scale: float = 0.01
    upsampling_ratios: Tuple[int, ...] = (2, 2)
    upsample_rates: Tuple[int, ...] = (8, 5, 4, 3)
    decoder_dim: int = 1536
    sample_rate: int = 24000

    @property
    def total_upsample(self) -> int:
        n = 1
        for r in tuple(self.upsampling_ratios) + tuple(self.upsample_rates):
            n *= r
        return n


@dataclass
class TTSConfig:
    talker: StackConfig = field(default_factory=StackConfig)
    predictor: StackConfig = field(
        default_factory=lambda: StackConfig(num_hidden_layers=5, vocab_size=2048)
    )
    codec: CodecConfig = field(default_factory=CodecConfig)
    num_code_groups: int = 16
    codec_eos_token_id: int = 2150
    codec_pad_id: int = 2148
    codec_bos_id: int = 2149
    codec_think_id: int = 2154
    codec_nothink_id: int = 2155
    codec_think_bos_id: int = 2156
    codec_think_eos_id: int = 2157
    codec_language_id: Dict[str, int] = field(default_factory=lambda: {"english": 2050, "chinese": 2055})
    spk_id: Dict[str, int] = field(default_factory=dict)
    spk_is_dialect: Dict[str, object] = field(default_factory=dict)
    text_vocab_size: int = 151936
    text_hidden_size: int = 2048
    tts_pad_token_id: int = 151671
    tts_bos_token_id: int = 151672
    tts_eos_token_id: int = 151673
    predictor_has_projection: bool = False   # small_to_mtp_projection is Identity when sizes match
    tts_model_type: str = "base"
    tts_model_size: str = "0b6"

    # ---- convenience -------------------------------------------------------
    @property
    def suppress_start(self) -> int:
        """First suppressed first-codebook id (``generate.py:47``)."""
        return max(0, self.talker.vocab_size - 1024)

    def to_dict(self) -> dict:
        return asdict(self)


def qwen3_tts_0p6b() -> TTSConfig:
    return TTSConfig()


def qwen3_tts_1p7b() -> TTSConfig:
    cfg = TTSConfig(
        talker=StackConfig(hidden_size=2048, intermediate_size=6144),
        predictor=StackConfig(num_hidden_layers=5, vocab_size=2048),
        predictor_has_projection=True,
        tts_model_size="1b7",
    )
    return cfg


def tiny_test_config(hidden: int = 256, layers: int = 2, pred_layers: int = 2,
                     heads: int = 4, kv_heads: int = 2, vocab: int = 1280,
                     pred_hidden: Optional[int] = None) -> TTSConfig:
    """Small shapes with the real head_dim (128) for CPU-speed parity tests."""
    ph = pred_hidden or hidden
    cfg = TTSConfig(
        talker=StackConfig(hidden_size=hidden, intermediate_size=hidden * 3, num_hidden_layers=layers,
                           num_attention_heads=heads, num_key_value_heads=kv_heads, vocab_size=vocab),
        predictor=StackConfig(hidden_size=ph, intermediate_size=ph * 3, num_hidden_layers=pred_layers,
                              num_attention_heads=heads, num_key_value_heads=kv_heads, vocab_size=256),
        codec=CodecConfig(codebook_size=256, codebook_dim=64, rvq_dim=32, latent_dim=128, hidden_size=64,
                          intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, head_dim=32,
                          sliding_window=8, decoder_dim=512),
        codec_eos_token_id=vocab - 1024 + 102,
        codec_pad_id=vocab - 1024 + 100, codec_bos_id=vocab - 1024 + 101,
        codec_think_id=vocab - 1024 + 106, codec_nothink_id=vocab - 1024 + 107,
        codec_think_bos_id=vocab - 1024 + 108, codec_think_eos_id=vocab - 1024 + 109,
        codec_language_id={"english": vocab - 1024 + 2, "chinese": vocab - 1024 + 7},
        text_vocab_size=512, text_hidden_size=hidden * 2,
        tts_pad_token_id=500, tts_bos_token_id=501, tts_eos_token_id=502,
        predictor_has_projection=(ph != hidden),
    )
    return cfg


def from_hf_config(d: dict) -> TTSConfig:
    """Map a Qwen3-TTS ``config.json`` dict (``talker_config`` / ``code_predictor_config`` /
    tokenizer ``decoder_config``) onto :class:`TTSConfig`.  Unknown keys keep defaults."""
    tc = d.get("talker_config", d)
    pc = tc.get("code_predictor_config", {})

    def stack(src: dict, base: StackConfig) -> StackConfig:
        kw = {}
        for k in ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                  "num_key_value_heads", "head_dim", "vocab_size", "rms_norm_eps"):
            if k in src:
                kw[k] = src[k]
        rp = src.get("rope_parameters") or {}
        if "rope_theta" in src:
            kw["rope_theta"] = float(src["rope_theta"])
        elif "rope_theta" in rp:
            kw["rope_theta"] = float(rp["rope_theta"])
        return StackConfig(**{**asdict(base), **kw})

    cfg = TTSConfig()
    cfg.talker = stack(tc, cfg.talker)
    cfg.predictor = stack(pc, cfg.predictor)
    for k in ("num_code_groups", "codec_eos_token_id", "codec_pad_id", "codec_bos_id", "codec_think_id",
              "codec_nothink_id", "codec_think_bos_id", "codec_think_eos_id", "codec_language_id",
              "spk_id", "spk_is_dialect", "text_vocab_size", "text_hidden_size"):
        if k in tc:
            setattr(cfg, k, tc[k])
    for k in ("tts_pad_token_id", "tts_bos_token_id", "tts_eos_token_id", "tts_model_type", "tts_model_size"):
        if k in d:
            setattr(cfg, k, d[k])
    cfg.predictor_has_projection = cfg.predictor.hidden_size != cfg.talker.hidden_size
    dc = d.get("decoder_config") or d.get("speech_tokenizer_decoder_config")
    if dc:
        base = asdict(cfg.codec)
        for k in base:
            if k in dc:
                base[k] = tuple(dc[k]) if isinstance(dc[k], list) else dc[k]
        cfg.codec = CodecConfig(**base)
    return cfg
