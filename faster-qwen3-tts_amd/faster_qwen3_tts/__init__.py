"""Drop-in import name: ``from faster_qwen3_tts import FasterQwen3TTS`` resolves to the MI355X HIP
implementation when ``faster-qwen3-tts_amd/`` is on ``sys.path`` (same module names as the reference
package: ``model``, ``generate``, ``streaming``, ``sampling``, ``talker_graph``, ``predictor_graph``)."""
import importlib
import sys

from fq3hip import __version__  # noqa: F401

for _m in ("model", "generate", "streaming", "sampling", "talker_graph", "predictor_graph", "cli"):
    sys.modules[f"{__name__}.{_m}"] = importlib.import_module(f"fq3hip.{_m}")


def __getattr__(name):
    if name == "FasterQwen3TTS":
        from fq3hip.model import FasterQwen3TTS
        return FasterQwen3TTS
    raise AttributeError(name)


__all__ = ["FasterQwen3TTS"]
