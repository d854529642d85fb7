"""fq3hip: MI355X-native fast decode path for Qwen3-TTS (HIP kernels behind a C ABI).

Importing the package is cheap and GPU-free; the HIP library is loaded on first use
(``fq3hip._lib.load``) and its absence is an ``ImportError``, never a silent fallback.
"""
from .config import TTSConfig, StackConfig, CodecConfig, qwen3_tts_0p6b, qwen3_tts_1p7b, tiny_test_config  # noqa: F401

__version__ = "0.1.0"
__all__ = ["FasterQwen3TTS", "TTSConfig", "StackConfig", "CodecConfig", "qwen3_tts_0p6b", "qwen3_tts_1p7b",
           "tiny_test_config"]


def __getattr__(name):
    if name == "FasterQwen3TTS":
        from .model import FasterQwen3TTS
        return FasterQwen3TTS
    raise AttributeError(name)
