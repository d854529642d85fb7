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
class RefAudioConfig:
    """Reference-audio analysers run once per new reference clip (reference ``model.py:430-447`` -> upstream
    ``create_voice_clone_prompt``): the speech tokenizer's ENCODER and the speaker encoder.

    Encoder defaults = the Mimi configuration (``transformers/models/mimi/configuration_mimi.py``: upstream's 12 Hz
    tokenizer encoder subclasses ``MimiModel`` and keeps the first 16 of its 32 quantizers) [recalled]; speaker defaults =
    the ECAPA-TDNN of ``modeling_qwen2_5_omni.py:2630-2707`` over a 128-bin log-mel (n_fft 1024, hop 256, 0-12 kHz,
    BigVGAN-style) [recalled].  ``enc_dim`` equals the talker's hidden size."""

    # speech-tokenizer encoder
    num_filters: int = 64
    ratios: Tuple[int, ...] = (4, 5, 6, 8)            # encoder order (= reversed Mimi ``upsampling_ratios``)
    kernel_size: int = 7
    last_kernel_size: int = 3
    residual_kernel_size: int = 3
    num_residual_layers: int = 1
    dilation_growth_rate: int = 2
    compress: int = 2
    hidden_size: int = 512
    num_hidden_layers: int = 8
    num_attention_heads: int = 8
    head_dim: int = 64
    intermediate_size: int = 2048
    sliding_window: int = 250
    norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    layer_scale_initial_scale: float = 0.01
    num_quantizers: int = 16                           # ``encoder_valid_num_quantizers``
    num_semantic_quantizers: int = 1
    codebook_size: int = 2048
    codebook_dim: int = 256
    max_positions: int = 8000                          # 25 Hz frames (320 s)
    sample_rate: int = 24000
    # speaker encoder
    mel_dim: int = 128
    n_fft: int = 1024
    hop_size: int = 256
    fmin: float = 0.0
    fmax: float = 12000.0
    enc_channels: Tuple[int, ...] = (512, 512, 512, 512, 1536)
    enc_kernel_sizes: Tuple[int, ...] = (5, 3, 3, 3, 1)
    enc_dilations: Tuple[int, ...] = (1, 2, 3, 4, 1)
    enc_attention_channels: int = 128
    enc_res2net_scale: int = 8
    enc_se_channels: int = 128
    enc_dim: int = 1024

    @property
    def n_bins_padded(self) -> int:
        return (self.n_fft // 2 + 1 + 31) // 32 * 32

    @property
    def samples_per_frame(self) -> int:
        n = 2
        for r in self.ratios:
            n *= r
        return n


def tiny_ref_audio_config() -> RefAudioConfig:
    """Small shapes (same structure) for CPU-speed oracle pins and quick GPU parity."""
    return RefAudioConfig(num_filters=64, ratios=(2, 3, 4), hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                          head_dim=32, intermediate_size=128, sliding_window=12, num_quantizers=4, codebook_size=256,
                          codebook_dim=32, max_positions=2048, mel_dim=32, n_fft=256, hop_size=64,
                          enc_channels=(256, 256, 256, 512), enc_kernel_sizes=(5, 3, 3, 1), enc_dilations=(1, 2, 3, 1),
                          enc_attention_channels=32, enc_res2net_scale=8, enc_se_channels=32, enc_dim=64)


@dataclass
class TTSConfig:
    talker: StackConfig = field(default_factory=StackConfig)
    predictor: StackConfig = field(
        default_factory=lambda: StackConfig(num_hidden_layers=5, vocab_size=2048)
    )
    codec: CodecConfig = field(default_factory=CodecConfig)
    ref_audio: RefAudioConfig = field(default_factory=RefAudioConfig)
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
        ref_audio=RefAudioConfig(enc_dim=2048),
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
    cfg.ref_audio = tiny_ref_audio_config()
    cfg.ref_audio.enc_dim = hidden
    cfg.ref_audio.num_quantizers = cfg.num_code_groups          # reference codes feed the 16-group prompt builder
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
    ra = asdict(cfg.ref_audio)
    ra["enc_dim"] = cfg.talker.hidden_size
    ec = d.get("encoder_config") or {}
    for k in ("num_filters", "kernel_size", "last_kernel_size", "residual_kernel_size", "dilation_growth_rate", "compress",
              "hidden_size", "num_hidden_layers", "num_attention_heads", "head_dim", "intermediate_size", "sliding_window",
              "norm_eps", "num_semantic_quantizers", "codebook_size", "codebook_dim", "layer_scale_initial_scale"):
        if ec.get(k) is not None:
            ra[k] = ec[k]
    if ec.get("num_residual_layers") is not None:
        ra["num_residual_layers"] = ec["num_residual_layers"]
    if ec.get("max_position_embeddings") is not None:
        ra["max_positions"] = int(ec["max_position_embeddings"])
    if ec.get("upsampling_ratios"):
        ra["ratios"] = tuple(reversed(ec["upsampling_ratios"]))
    if d.get("encoder_valid_num_quantizers"):
        ra["num_quantizers"] = int(d["encoder_valid_num_quantizers"])
    sc = d.get("speaker_encoder_config") or {}
    # the mel front end's parameters are constants in upstream's code; accepted here when a config carries them
    for k in ("mel_dim", "enc_dim", "enc_attention_channels", "enc_res2net_scale", "enc_se_channels", "n_fft", "hop_size", "fmin", "fmax"):
        if sc.get(k) is not None:
            ra[k] = sc[k]
    for k in ("enc_channels", "enc_kernel_sizes", "enc_dilations"):
        if sc.get(k):
            ra[k] = tuple(sc[k])
    cfg.ref_audio = RefAudioConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in ra.items()})
    return cfg
