#!/usr/bin/env python3
"""Golden prompt embeddings produced by the REFERENCE's own prompt builder: imports
/root/reference/faster_qwen3_tts/model.py (with a `soundfile` stub) and calls
``FasterQwen3TTS._build_talker_inputs_local(None, m, ...)`` (model.py:583-805) over the CPU duck-typed model of
oracle/qwen3tts_oracle.py::OraclePromptModel, on the tiny config with seeded weights.

    python oracle/make_golden_prompt.py        # -> tests/golden/prompt.npz  (needs /root/reference; build container only)

Cases cover every branch of Appendix B: x-vector streaming-text, x-vector non-streaming-text with language "auto", ICL with
the text stream longer / shorter than the codec stream, ICL non-streaming, custom-voice speaker id with an instruct turn.
Test infrastructure only.
"""
import copy
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

from fq3hip.config import tiny_test_config            # noqa: E402
from fq3hip.weights import synth_weights              # noqa: E402
from oracle import qwen3tts_oracle as O               # noqa: E402


def prompt_cases(cfg):
    """(name, kwargs) -- deterministic token ids; shared with tests/test_gpu_prompt.py."""
    g = torch.Generator().manual_seed(31)
    ids = lambda n: torch.randint(16, cfg.text_vocab_size - 16, (1, n), generator=g)
    H = cfg.talker.hidden_size
    spk = torch.randn(H, generator=g)
    code = lambda n: torch.cat([torch.randint(0, cfg.talker.vocab_size - 1024, (n, 1), generator=g),
                                torch.randint(0, cfg.predictor.vocab_size, (n, cfg.num_code_groups - 1), generator=g)], 1)
    xv = dict(ref_code=[None], ref_spk_embedding=[spk], x_vector_only_mode=[True], icl_mode=[False])
    icl = lambda n: dict(ref_code=[code(n)], ref_spk_embedding=[spk], x_vector_only_mode=[False], icl_mode=[True])
    return [
        ("xvec_stream", dict(input_id=ids(3 + 9 + 5), ref_id=None, vcp=xv, language="English", speaker=None, nsm=False, instruct=None)),
        ("xvec_nonstream_auto", dict(input_id=ids(3 + 6 + 5), ref_id=None, vcp=xv, language="Auto", speaker=None, nsm=True, instruct=None)),
        ("icl_text_longer", dict(input_id=ids(3 + 14 + 5), ref_id=ids(3 + 6 + 2), vcp=icl(7), language="English", speaker=None, nsm=False, instruct=None)),
        ("icl_codec_longer", dict(input_id=ids(3 + 4 + 5), ref_id=ids(3 + 3 + 2), vcp=icl(19), language="Chinese", speaker=None, nsm=False, instruct=None)),
        ("icl_nonstream", dict(input_id=ids(3 + 5 + 5), ref_id=ids(3 + 4 + 2), vcp=icl(6), language="English", speaker=None, nsm=True, instruct=None)),
        ("custom_voice_instruct", dict(input_id=ids(3 + 8 + 5), ref_id=None, vcp=None, language="English", speaker="bob", nsm=True, instruct=ids(7))),
    ]


def case_config():
    cfg = tiny_test_config()
    cfg = copy.deepcopy(cfg)
    cfg.spk_id = {"bob": cfg.talker.vocab_size - 1024 + 300}
    cfg.spk_is_dialect = {"bob": False}
    return cfg


def main():
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
    sys.path.insert(0, "/root/reference")
    from faster_qwen3_tts.model import FasterQwen3TTS as RefTTS
    cfg = case_config()
    out = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor", "text"))
        m = O.OraclePromptModel(cfg, W)
        for name, c in prompt_cases(cfg):
            vcp = c["vcp"]
            if vcp is not None:
                vcp = dict(vcp, ref_spk_embedding=[e.to(dtype) for e in vcp["ref_spk_embedding"]])
            with torch.inference_mode():
                tie, tam, tth, tpe = RefTTS._build_talker_inputs_local(
                    None, m, [c["input_id"]], [c["ref_id"]], vcp, [c["language"]], [c["speaker"]], c["nsm"], [c["instruct"]])
            assert int(tam.sum()) == tie.shape[1]
            for k, v in (("tie", tie), ("tth", tth), ("tpe", tpe)):
                out[f"{name}_{tag}_{k}"] = v.float().numpy()
            print(f"{name} {tag}: L={tie.shape[1]} trailing={tth.shape[1]}")
    path = os.path.join(ROOT, "tests", "golden", "prompt.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
