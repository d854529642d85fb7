"""GPU: the HIP prompt builder (fq3_text_project + fq3_prompt_rows behind FasterQwen3TTS._build_talker_inputs_local) vs
goldens produced by the REFERENCE's own prompt builder on CPU (oracle/make_golden_prompt.py -> tests/golden/prompt.npz).
fp32: <= 1e-4 of the tensor scale (summation order inside the MLP only); bf16: <= 2 bf16 ulps of the tensor scale with at
least 90 % of the elements bit-equal (a 512-long dot product in another order may round a sum the other way)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fq3hip.weights import synth_weights
from oracle.make_golden_prompt import prompt_cases, case_config


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_hip_prompt_matches_reference_builder(tag, dtype, golden_dir):
    from fq3hip.model import FasterQwen3TTS
    g = np.load(os.path.join(golden_dir, "prompt.npz"))
    cfg = case_config()
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor", "text"))
    model = FasterQwen3TTS.from_weights(cfg, W, device="cuda", dtype=dtype, max_seq_len=96, max_frames=32)
    m = model.model.model
    assert m.talker.hip_prompt_ready
    for name, c in prompt_cases(cfg):
        vcp = c["vcp"]
        if vcp is not None:
            vcp = dict(vcp, ref_spk_embedding=[e.to(dtype) for e in vcp["ref_spk_embedding"]])
        tie, tam, tth, tpe = model._build_talker_inputs_local(m, [c["input_id"].cuda()], [c["ref_id"].cuda() if c["ref_id"] is not None else None],
                                                              vcp, [c["language"]], [c["speaker"]], c["nsm"],
                                                              [c["instruct"].cuda() if c["instruct"] is not None else None])
        for k, v in (("tie", tie), ("tth", tth), ("tpe", tpe)):
            ref = torch.from_numpy(g[f"{name}_{tag}_{k}"])
            got = v.float().cpu()
            assert got.shape == ref.shape, (name, k)
            scale = max(1.0, float(ref.abs().max()))
            err = float((got - ref).abs().max())
            if dtype == torch.float32:
                assert err <= 1e-4 * scale, (name, k, err)
            else:
                assert err <= 2.0 ** -6 * scale, (name, k, err)
                assert float((got == ref).float().mean()) >= 0.9, (name, k)
        assert int(tam.sum()) == tie.shape[1] and tam.shape == (1, tie.shape[1])


def test_generic_tensor_path_agrees_with_hip_path():
    """The tensor-op builder kept for foreign model objects gives the same prompt as the HIP builder (fp32)."""
    from fq3hip.model import FasterQwen3TTS
    cfg = case_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("talker", "predictor", "text"))
    model = FasterQwen3TTS.from_weights(cfg, W, device="cuda", dtype=torch.float32, max_seq_len=96, max_frames=32)
    m = model.model.model
    for name, c in prompt_cases(cfg):
        args = ([c["input_id"].cuda()], [c["ref_id"].cuda() if c["ref_id"] is not None else None], c["vcp"], [c["language"]],
                [c["speaker"]], c["nsm"], [c["instruct"].cuda() if c["instruct"] is not None else None])
        a = model._build_talker_inputs_local(m, *args)
        m.talker.hip_prompt_ready = False
        try:
            b = model._build_talker_inputs_local(m, *args)
        finally:
            m.talker.hip_prompt_ready = True
        for x, y in zip(a, b):
            assert x.shape == y.shape and float((x.float() - y.float()).abs().max()) <= 1e-4 * max(1.0, float(y.float().abs().max())), name
