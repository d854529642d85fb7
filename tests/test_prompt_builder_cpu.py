"""CPU: the HOST logic of the HIP prompt builder (fq3hip/prompt.py: which token / codec id / reference frame goes to which
row) against goldens produced by the REFERENCE's own ``_build_talker_inputs_local`` (oracle/make_golden_prompt.py).  The
two device entry points are emulated here with plain tensor ops, so every row program is checked without a GPU; the GPU
test (tests/test_gpu_prompt.py) then checks the kernels behind the same programs."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fq3hip.prompt import build_talker_inputs_hip
from fq3hip.weights import synth_weights
from oracle import qwen3tts_oracle as O
from oracle.make_golden_prompt import prompt_cases, case_config


class _FakeEngine:
    """tensor-op stand-in for Fq3Engine.text_project / prompt_rows (test infrastructure)."""

    def __init__(self, cfg, W):
        self.cfg, self.W, self.device = cfg, W, torch.device("cpu")
        self.tabs = [W["talker.model.codec_embedding.weight"]] + [W[f"talker.code_predictor.model.codec_embedding.{j}.weight"]
                                                                  for j in range(cfg.num_code_groups - 1)]

    def text_project(self, ids):
        W = self.W
        x = F.embedding(ids.reshape(-1), W["talker.model.text_embedding.weight"])
        h = F.silu(F.linear(x, W["talker.text_projection.linear_fc1.weight"], W["talker.text_projection.linear_fc1.bias"]))
        return F.linear(h, W["talker.text_projection.linear_fc2.weight"], W["talker.text_projection.linear_fc2.bias"])

    def prompt_rows(self, text_rows, prog, ref_codes=None, spk_embed=None):
        out = []
        for trow, kind, arg in prog.tolist():
            c = None
            if kind == 1:
                c = self.tabs[0][arg]
            elif kind == 2:
                c = spk_embed.reshape(-1).to(self.tabs[0].dtype)
            elif kind == 3:
                c = torch.stack([self.tabs[g][int(ref_codes[arg, g])] for g in range(16)]).sum(0)
            t = text_rows[trow] if trow >= 0 else None
            out.append(t + c if (t is not None and c is not None) else (t if t is not None else c))
        return torch.stack(out)


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_row_programs_reproduce_the_reference_prompt(tag, dtype, golden_dir):
    g = np.load(os.path.join(golden_dir, "prompt.npz"))
    cfg = case_config()
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor", "text"))
    om = O.OraclePromptModel(cfg, W)
    m = NS(talker=NS(engine=_FakeEngine(cfg, W)), config=om.config, generate_speaker_prompt=om.generate_speaker_prompt)
    for name, c in prompt_cases(cfg):
        vcp = c["vcp"]
        if vcp is not None:
            vcp = dict(vcp, ref_spk_embedding=[e.to(dtype) for e in vcp["ref_spk_embedding"]])
        tie, tam, tth, tpe = build_talker_inputs_hip(m, c["input_id"], c["ref_id"], vcp, 0, c["language"], c["speaker"], c["nsm"],
                                                     c["instruct"])
        for k, v in (("tie", tie), ("tth", tth), ("tpe", tpe)):
            ref = torch.from_numpy(g[f"{name}_{tag}_{k}"])
            assert v.shape == ref.shape, (name, k, v.shape, ref.shape)
            # same tensor ops on the same CPU: the only freedom is the batching of the MLP rows, which F.linear does not
            # guarantee to be bit-stable in bf16 -> compare to 1 bf16 ulp there, exactly-ish in fp32
            tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
            assert (v.float() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max())), (name, k)
        assert int(tam.sum()) == tie.shape[1]


def test_host_ids_note_is_used_and_voided_by_a_write():
    """prompt.host_ids (round 6): the tokenising step notes the id list it uploaded on the tensor, the prompt builder reads the note
    instead of copying the tensor back -- and a tensor written to since (or from anywhere else) is read back."""
    from fq3hip.prompt import host_ids
    t = torch.tensor([[5, 6, 7]], dtype=torch.long)
    assert host_ids(t) == [5, 6, 7]                                   # no note: read back
    t.fq3_host_ids = ([5, 6, 7], t._version)
    t_data = t.clone()
    assert host_ids(t) == [5, 6, 7]
    t.fq3_host_ids = ([9, 9, 9], t._version)                          # (a note that differs from the data proves the note is what is read)
    assert host_ids(t) == [9, 9, 9]
    t[0, 0] = 1                                                       # an in-place write bumps the version: the note is void
    assert host_ids(t) == [1, 6, 7]
    with torch.inference_mode():
        u = torch.tensor([[2, 3]], dtype=torch.long)
    u.fq3_host_ids = ([2, 3], None)                                   # inference-mode tensors carry no version counter: taken as noted
    assert host_ids(u) == [2, 3]
    assert t_data.tolist() == [[5, 6, 7]]


def test_tokenize_texts_notes_the_ids_it_uploads():
    from fq3hip.native_model import ByteTokenizer, NativeQwen3TTS
    from fq3hip.prompt import host_ids
    m = NativeQwen3TTS.__new__(NativeQwen3TTS)
    m.tokenizer, m.device = ByteTokenizer(4096), torch.device("cpu")
    (t,) = m._tokenize_texts(["<|im_start|>assistant\nhello<|im_end|>\n<|im_start|>assistant\n"])
    assert isinstance(getattr(t, "fq3_host_ids", None), tuple)
    assert host_ids(t) == [int(x) for x in t.reshape(-1).tolist()]
