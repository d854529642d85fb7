"""CPU: the oracle against the committed golden vectors (which oracle/make_golden.py produced from the
reference's own sampling.py / generate.py / streaming.py and the transformers sibling modules)."""
import os

import numpy as np
import pytest
import torch

from fq3hip.config import tiny_test_config
from fq3hip.weights import synth_weights, synth_prompt
from oracle import qwen3tts_oracle as O


def test_sampler_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "sampler.npz"), allow_pickle=True)
    assert len(g["cases"]) >= 100
    for case in g["cases"]:
        dt = torch.bfloat16 if case["bf16"] else torch.float32
        V = int(case["V"])
        logits = torch.from_numpy(case["logits"]).to(dt).view(1, V)
        noise = torch.from_numpy(case["noise"]).to(dt).view(1, V)
        sm = O.build_suppress_mask(V, int(case["eos"]))
        sup = [int(case["eos"])] if case["sup_eos"] else None
        tok = O.sample_logits(logits, suppress_mask=sm, suppress_tokens=sup, noise=noise,
                              temperature=float(case["temperature"]), top_k=int(case["top_k"]),
                              top_p=float(case["top_p"]), do_sample=bool(case["do_sample"]))
        assert int(tok) == int(case["token"])


def test_repetition_penalty_kat():
    # reference tests/test_sampling.py:10-21
    logits = torch.zeros(1, 1, 10)
    logits[..., 7] = 1.0
    logits[..., 8] = -1.0
    others = [0, 1, 2, 3, 4, 5, 6, 8, 9]
    hist = torch.tensor([7] + [others[i % len(others)] for i in range(1, 60)])
    out = O.apply_repetition_penalty(logits.clone(), hist, 1.1)
    assert out[0, 0, 7].item() == pytest.approx(1.0 / 1.1, rel=1e-6)
    assert out[0, 0, 8].item() == pytest.approx(-1.1, rel=1e-6)


@pytest.mark.parametrize("tag,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_decode_codes_and_chunks(tag, dt, golden_dir):
    g = np.load(os.path.join(golden_dir, "decode.npz"))
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, dt)
    for case in range(3):
        plen, tlen, maxnew, minnew, rp = g[f"params_{tag}_{case}"]
        tie, tam, tth, tpe, _ = synth_prompt(cfg, int(plen), int(tlen), 0, dtype=dt)
        orc = O.OracleTTS(cfg, W, max_seq_len=96)
        orc.pred_sampling = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
        sp = O.SamplingParams(max_new_tokens=int(maxnew), min_new_tokens=int(minnew), temperature=1.0, top_k=0,
                              top_p=1.0, do_sample=False, repetition_penalty=float(rp))
        codes = orc.generate(tie, tam, tth, tpe, sp)
        ref = torch.from_numpy(g[f"codes_{tag}_{case}"])
        assert torch.equal(codes, ref)
        # structural validity, reference tests/test_e2e_parity.py:40-101
        assert codes.shape[1] == 16 and (codes[:, 0] < cfg.talker.vocab_size - 1024).all()
        assert (codes[:, 0] != cfg.codec_eos_token_id).all()
        metas = [(m["chunk_index"], m["chunk_steps"], m["total_steps_so_far"]) for _, m in O.stream_chunks(codes, 8)]
        assert metas == [tuple(r[:3]) for r in g[f"chunks_{tag}_{case}"].tolist()]


def test_stack_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "stack.npz"))
    cfg = tiny_test_config()
    pc = cfg.predictor
    W = synth_weights(cfg, 0, torch.float32)
    x, y = torch.from_numpy(g["x_f32"]), torch.from_numpy(g["y_f32"])
    cache = O.KVCache.empty(pc.num_hidden_layers, 17, pc.num_key_value_heads, pc.head_dim, torch.float32)
    h = O.stack_forward(W, "talker.code_predictor.model", pc, x[:2], 0, cache, torch.arange(2).float())
    assert (h - y[:2]).abs().max() < 1e-5
    for i in range(2, x.shape[0]):
        h = O.stack_forward(W, "talker.code_predictor.model", pc, x[i:i + 1], i, cache, torch.tensor([float(i)]))
        assert (h - y[i:i + 1]).abs().max() < 1e-5


def test_codec_vectors_and_lengths(golden_dir):
    g = np.load(os.path.join(golden_dir, "codec.npz"))
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("codec",))
    wav = O.codec_decode(torch.from_numpy(g["codes"]), W, cfg.codec).numpy()
    assert np.abs(wav - g["wav_f32"]).max() < 1e-4      # thread-count dependent fp32 summation order
    # sample count law of the causal transposed convs (k=2r trims r on both sides)
    n = g["codes"].shape[0]
    for f in cfg.codec.upsampling_ratios:
        n *= f
    for r in cfg.codec.upsample_rates:
        n = (n - 1) * r
    assert wav.shape[0] == n


def test_streaming_vocoder_windowing_matches_full_decode_early():
    """model.py:1085-1115: phase-1 chunks concatenate to exactly the full decode of the same frames."""
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("codec",))
    tok = O.OracleSpeechTokenizer(cfg, W)
    g = torch.Generator().manual_seed(1)
    codes = torch.randint(0, cfg.codec.codebook_size, (24, 16), generator=g)
    chunks = [codes[i:i + 8] for i in range(0, 24, 8)]
    parts = list(O.streaming_vocode(tok, chunks, None, 8))
    full = tok.decode({"audio_codes": codes.unsqueeze(0)})[0][0].numpy()
    assert np.abs(np.concatenate(parts) - full).max() < 2e-4


def test_fulldepth_golden_is_the_oracle(golden_dir):
    """tests/golden/fulldepth.npz (the ids the GPU full-depth parity tests are scored against) is reproduced by the
    oracle: regenerate the first frames of the 0.6B fp32 case at full depth and compare ids and margins."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(golden_dir), "..", "oracle"))
    from oracle import make_golden_fulldepth as M
    g = np.load(os.path.join(golden_dir, "fulldepth.npz"))
    r, _ = M.run_case("0p6b", torch.float32, frames=3)
    assert np.array_equal(r["codes"], g["0p6b_f32_codes"][:3])
    assert np.allclose(r["t_margin"][:3], g["0p6b_f32_t_margin"][:3], atol=1e-4)
    assert np.allclose(r["p_top1"], g["0p6b_f32_p_top1"][:3], atol=1e-4)


@pytest.mark.parametrize("sizes,ref_frames", [([8] * 6, 5), ([4, 4, 4, 4, 4, 4, 4, 4, 4, 3], 0), ([12, 12, 12, 7], 9), ([8, 16, 8, 8, 3], 4)])
def test_product_streaming_vocoder_is_the_reference_windowing(sizes, ref_frames):
    """fq3hip.model.StreamingVocoder (the state machine behind generate_voice_clone_streaming AND the batched streaming
    paths, here on its synchronous foreign-tokenizer path) emits exactly what the restated reference windowing
    (model.py:1052-1137) emits -- also for ragged chunk sizes, which only the batched scheduler produces."""
    from fq3hip.model import StreamingVocoder
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("codec",))
    tok = O.OracleSpeechTokenizer(cfg, W)
    g = torch.Generator().manual_seed(3)
    n = sum(sizes)
    codes = torch.randint(0, cfg.codec.codebook_size, (n, 16), generator=g)
    ref = torch.randint(0, cfg.codec.codebook_size, (ref_frames, 16), generator=g) if ref_frames else None
    chunks, pos = [], 0
    for k in sizes:
        chunks.append(codes[pos:pos + k]); pos += k
    want = list(O.streaming_vocode(tok, chunks, ref, sizes[0]))
    voc = StreamingVocoder(tok, ref, sizes[0], "cpu", side_stream=None)
    got = [voc.push(c)[0] for c in chunks]
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert voc.spf is not None or n < max(25, sizes[0])
