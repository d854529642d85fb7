"""GPU: the public API end to end on a synthetic-weight model (text -> prompt -> prefill -> hipGraph
decode -> HIP vocoder), checked against the CPU oracle run on the very prompt embeddings the wrapper
built.  Relations taken from the reference's tests/test_e2e_parity.py: exact greedy ids in fp32
(:431-485 x-vector, :488-582 ICL), streaming == non-streaming ids (:729-782), structural validity."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fq3hip.config import tiny_test_config
from fq3hip.weights import synth_weights

GREEDY = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0, repetition_penalty=1.0, min_new_tokens=0)


@pytest.fixture(scope="module")
def model_and_weights():
    from fq3hip.model import FasterQwen3TTS
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("talker", "predictor", "codec", "text"))
    m = FasterQwen3TTS.from_weights(cfg, W, device="cuda", dtype=torch.float32, max_seq_len=160, codec_max_frames=128,
                                    max_frames=64)
    m.predictor_graph.do_sample = False       # tests/test_e2e_parity.py:208-215: force greedy before warmup
    m.predictor_graph.top_k = 0
    return cfg, W, m


def _oracle_codes(cfg, W, prep, max_new):
    from oracle import qwen3tts_oracle as O
    _, _, _, tie, tam, tth, tpe, _ = prep
    orc = O.OracleTTS(cfg, W, max_seq_len=160)
    orc.pred_sampling = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    sp = O.SamplingParams(max_new_tokens=max_new, **GREEDY)
    return orc.generate(tie.cpu(), tam.cpu(), tth.cpu(), tpe.cpu(), sp)


@pytest.mark.parametrize("mode", ["xvec", "icl", "icl_nonstreaming_text"])
def test_voice_clone_matches_oracle(model_and_weights, mode):
    cfg, W, m = model_and_weights
    g = torch.Generator().manual_seed(4)
    spk = torch.randn(cfg.talker.hidden_size, generator=g)
    if mode == "xvec":
        vcp = dict(ref_spk_embedding=[spk])
        kw = dict(ref_text="")
    else:
        ref = torch.randint(0, cfg.codec.codebook_size, (12, 16), generator=g)
        vcp = dict(ref_spk_embedding=[spk], ref_code=[ref], x_vector_only_mode=[False], icl_mode=[True])
        kw = dict(ref_text="the reference sentence")
    nsm = mode == "icl_nonstreaming_text"
    text = "A short line to speak."
    prep = m._prepare_generation(text=text, language="English", voice_clone_prompt=vcp, non_streaming_mode=nsm, **kw)
    ref_codes = _oracle_codes(cfg, W, prep, 24)
    from fq3hip.generate import fast_generate
    _, talker, config, tie, tam, tth, tpe, rc = prep
    codes, timing = fast_generate(talker, tie, tam, tth, tpe, config, m.predictor_graph, m.talker_graph,
                                  max_new_tokens=24, **GREEDY)
    assert torch.equal(codes.cpu(), ref_codes)
    assert set(timing) == {"prefill_ms", "decode_s", "steps", "ms_per_step", "steps_per_s"}
    # structural validity (tests/test_e2e_parity.py:40-101)
    assert codes.shape[1] == 16 and int(codes[:, 0].max()) < cfg.talker.vocab_size - 1024
    assert (codes[:, 0] != cfg.codec_eos_token_id).all()
    # public call: audio comes back as float numpy at 24 kHz, reference part trimmed in ICL mode
    audio, sr = m.generate_voice_clone(text, "English", voice_clone_prompt=vcp, max_new_tokens=24,
                                       non_streaming_mode=nsm, **GREEDY, **kw)
    assert sr == 24000 and isinstance(audio[0], np.ndarray) and audio[0].dtype == np.float32
    tok = m.speech_tokenizer
    n_total = codes.shape[0] + (12 if mode != "xvec" else 0)
    full = tok.num_samples(n_total)
    expect = full - int((12 if mode != "xvec" else 0) / n_total * full)
    assert len(audio[0]) == expect and np.isfinite(audio[0]).all() and np.abs(audio[0]).max() <= 1.0


def test_streaming_equals_non_streaming(model_and_weights):
    cfg, W, m = model_and_weights
    from fq3hip.generate import fast_generate
    from fq3hip.streaming import fast_generate_streaming
    g = torch.Generator().manual_seed(5)
    vcp = dict(ref_spk_embedding=[torch.randn(cfg.talker.hidden_size, generator=g)])
    _, talker, config, tie, tam, tth, tpe, _ = m._prepare_generation(text="Streaming parity.", language="Auto", voice_clone_prompt=vcp)
    codes, _ = fast_generate(talker, tie, tam, tth, tpe, config, m.predictor_graph, m.talker_graph, max_new_tokens=21, **GREEDY)
    chunks = list(fast_generate_streaming(talker, tie, tam, tth, tpe, config, m.predictor_graph, m.talker_graph,
                                          max_new_tokens=21, chunk_size=8, **GREEDY))
    assert torch.equal(torch.cat([c for c, _ in chunks]), codes)
    metas = [(t["chunk_index"], t["chunk_steps"], t["total_steps_so_far"], t["is_final"]) for _, t in chunks]
    assert metas == [(0, 8, 8, False), (1, 8, 16, False), (2, 5, 21, True)]
    assert chunks[0][1]["prefill_ms"] > 0 and chunks[1][1]["prefill_ms"] == 0
    # the reference's keys, plus one extra (an event that lets another stream wait for just this chunk)
    assert set(chunks[0][1]) - {"codes_ready_event"} == {"chunk_index", "chunk_steps", "prefill_ms", "decode_ms",
                                                          "total_steps_so_far", "is_final"}
    # parity_mode (no hipGraph) gives the same ids
    codes2, _ = fast_generate(talker, tie, tam, tth, tpe, config, m.predictor_graph, m.talker_graph, max_new_tokens=21,
                              parity_mode=True, **GREEDY)
    assert torch.equal(codes2, codes)


def test_streaming_audio_matches_oracle_windowing(model_and_weights):
    """generate_voice_clone_streaming's phase-1/phase-2 windowing vs the oracle's restatement of
    model.py:1052-1137 over the oracle vocoder (fp32: PCM RMS <= 1e-3)."""
    from oracle import qwen3tts_oracle as O
    cfg, W, m = model_and_weights
    g = torch.Generator().manual_seed(6)
    ref = torch.randint(0, cfg.codec.codebook_size, (10, 16), generator=g)
    vcp = dict(ref_spk_embedding=[torch.randn(cfg.talker.hidden_size, generator=g)], ref_code=[ref],
               x_vector_only_mode=[False], icl_mode=[True])
    outs = list(m.generate_voice_clone_streaming("Windowed decode check, long enough.", "English", ref_text="ref words",
                                                 voice_clone_prompt=vcp, max_new_tokens=40, chunk_size=8,
                                                 **{**GREEDY, "min_new_tokens": 40}))
    assert len(outs) == 5 and all(sr == 24000 for _, sr, _ in outs)
    prep = m._prepare_generation(text="Windowed decode check, long enough.", language="English", ref_text="ref words",
                                 voice_clone_prompt=vcp)
    from fq3hip.generate import fast_generate
    _, talker, config, tie, tam, tth, tpe, rc = prep
    codes, _ = fast_generate(talker, tie, tam, tth, tpe, config, m.predictor_graph, m.talker_graph, max_new_tokens=40,
                             **{**GREEDY, "min_new_tokens": 40})
    codes = codes.cpu()
    tok = O.OracleSpeechTokenizer(cfg, W)
    want = list(O.streaming_vocode(tok, [codes[i:i + 8] for i in range(0, 40, 8)], ref, 8))
    for (a, _, _), b in zip(outs, want):
        assert a.shape == b.shape
        assert float(np.sqrt(np.mean((a - b) ** 2))) <= 1e-3


def test_custom_voice_requires_matching_model_type(model_and_weights):
    _, _, m = model_and_weights
    with pytest.raises(ValueError, match="does not support custom voice"):
        m.generate_custom_voice("hi", "bob", "English")
    with pytest.raises(ValueError, match="does not support voice design"):
        m.generate_voice_design("hi", "a calm voice", "English")


def test_sampling_module_on_gpu():
    from fq3hip.sampling import sample_logits, apply_repetition_penalty
    logits = torch.randn(1, 3072, device="cuda")
    mask = torch.zeros(3072, dtype=torch.bool, device="cuda")
    mask[2048:] = True
    tok = sample_logits(logits, temperature=1.0, top_k=0, top_p=1.0, do_sample=False, suppress_mask=mask)
    assert int(tok) == int(torch.argmax(logits[0, :2048]))
    l2 = torch.zeros(1, 1, 10, device="cuda"); l2[..., 7] = 1.0; l2[..., 8] = -1.0
    out = apply_repetition_penalty(l2.clone(), torch.tensor([7, 8, 8], device="cuda"), 1.1)
    assert out[0, 0, 7].item() == pytest.approx(1 / 1.1, rel=1e-6) and out[0, 0, 8].item() == pytest.approx(-1.1, rel=1e-6)


def test_prefill_kv_import_from_hf_style_cache(model_and_weights):
    """TalkerGraph.prefill_kv with an HF-style cache (layer -> (k, v) [1, kv_heads, L, d]), talker_graph.py:153-170."""
    from fq3hip.engine import Fq3Engine
    from fq3hip.talker_graph import TalkerGraph
    cfg, W, m = model_and_weights
    eng = m.talker_graph.engine
    g = torch.Generator().manual_seed(8)
    x = (torch.randn(37, cfg.talker.hidden_size, generator=g) * 0.5).cuda()
    eng.prefill(x.contiguous())
    cache = []
    for li in range(cfg.talker.num_hidden_layers):
        k, v = eng.kv_export(li, 37)
        cache.append((k.unsqueeze(0).clone(), v.unsqueeze(0).clone()))
    step_in = torch.randn(cfg.talker.hidden_size, generator=g).cuda()
    want = eng.talker_step(step_in, 37).clone()
    eng2 = Fq3Engine(cfg, W, device="cuda", dtype=torch.float32, max_seq_len=64, max_frames=8)
    tg2 = TalkerGraph(eng2)
    assert tg2.prefill_kv(cache) == 37
    tg2.set_generation_state(torch.ones(1, 37, dtype=torch.long), None)
    got = tg2.run(step_in.view(1, 1, -1), 37)
    assert torch.equal(got.view(-1), want)
    tiny = Fq3Engine(cfg, W, device="cuda", dtype=torch.float32, max_seq_len=32, max_frames=8)
    with pytest.raises(RuntimeError, match="Input is too long"):
        TalkerGraph(tiny).prefill_kv(cache)
    # and against the ORACLE: a cache computed on the CPU by the oracle's prefill (the upstream talker.forward the
    # reference hands to prefill_kv, generate.py:107-137) is imported, then one decode step must equal the oracle's
    from oracle import qwen3tts_oracle as O
    orc = O.OracleTTS(cfg, {k: v.float().cpu() for k, v in W.items()}, max_seq_len=64)
    orc.prefill(x.cpu().unsqueeze(0), torch.ones(1, 37, dtype=torch.long))
    ocache = [(orc.tcache.k[li][:37].permute(1, 0, 2).unsqueeze(0).contiguous(), orc.tcache.v[li][:37].permute(1, 0, 2).unsqueeze(0).contiguous())
              for li in range(cfg.talker.num_hidden_layers)]
    eng3 = Fq3Engine(cfg, W, device="cuda", dtype=torch.float32, max_seq_len=64, max_frames=8)
    tg3 = TalkerGraph(eng3)
    assert tg3.prefill_kv([(k.cuda(), v.cuda()) for k, v in ocache]) == 37
    tg3.set_generation_state(torch.ones(1, 37, dtype=torch.long), None)
    got3 = tg3.run(step_in.view(1, 1, -1), 37).view(-1).float().cpu()
    ref3 = orc.talker_step(step_in.cpu().view(1, 1, -1), 37).view(-1)
    assert (got3 - ref3).abs().max() <= 2e-4 * max(1.0, float(ref3.abs().max()))


@pytest.mark.parametrize("kind", ["custom_voice", "voice_design"])
def test_custom_voice_and_voice_design_paths(kind):
    """generate_custom_voice / generate_voice_design (model.py:1139-1505): same decode core, different prompt."""
    import copy
    from fq3hip.model import FasterQwen3TTS
    from fq3hip.generate import fast_generate
    cfg = copy.deepcopy(tiny_test_config())
    cfg.tts_model_type = kind
    cfg.tts_model_size = "1b7"
    cfg.spk_id = {"bob": 7}
    cfg.spk_is_dialect = {"bob": False}
    W = synth_weights(cfg, 0, torch.float32, parts=("talker", "predictor", "codec", "text"))
    m = FasterQwen3TTS.from_weights(cfg, W, device="cuda", dtype=torch.float32, max_seq_len=160, codec_max_frames=64, max_frames=32)
    m.predictor_graph.do_sample = False
    m.predictor_graph.top_k = 0
    text = "Say this in a designed voice."
    if kind == "custom_voice":
        prep = m._custom_prepare(text, "bob", "English", "speak slowly", None)
        with pytest.raises(ValueError):
            m.generate_custom_voice(text, "alice", "English")
        with pytest.raises(ValueError):
            m.generate_custom_voice(text, "bob", "Klingon")
    else:
        prep = m._design_prepare(text, "a calm low voice", "English", None)
    _, talker, config, tie, tam, tth, tpe = prep
    assert tth.shape[1] == 1            # non_streaming_mode defaults to True here: the whole text is prefilled
    codes, _ = fast_generate(talker, tie, tam, tth, tpe, config, m.predictor_graph, m.talker_graph, max_new_tokens=10, **GREEDY)
    from oracle import qwen3tts_oracle as O
    orc = O.OracleTTS(cfg, W, max_seq_len=160)
    orc.pred_sampling = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    ref = orc.generate(tie.cpu(), tam.cpu(), tth.cpu(), tpe.cpu(), O.SamplingParams(max_new_tokens=10, **GREEDY))
    assert torch.equal(codes.cpu(), ref)
    if kind == "custom_voice":
        audio, sr = m.generate_custom_voice(text, "bob", "English", instruct="speak slowly", max_new_tokens=10, **GREEDY)
        chunks = list(m.generate_custom_voice_streaming(text, "bob", "English", max_new_tokens=10, chunk_size=4, **GREEDY))
    else:
        audio, sr = m.generate_voice_design(text, "a calm low voice", "English", max_new_tokens=10, **GREEDY)
        chunks = list(m.generate_voice_design_streaming(text, "a calm low voice", "English", max_new_tokens=10, chunk_size=4, **GREEDY))
    assert sr == 24000 and len(audio[0]) == m.speech_tokenizer.num_samples(codes.shape[0])
    assert sum(c[2]["chunk_steps"] for c in chunks) == codes.shape[0]
