#!/usr/bin/env python3
"""Generate and verify the golden vectors under ``tests/golden/`` (ORACLE-side tooling).

Runs ONLY in the build container, where ``/root/reference`` is mounted read-only.  It imports the
reference's own Python (``faster_qwen3_tts/sampling.py``, ``generate.py``, ``streaming.py``) and the
transformers-5.15 sibling modules that carry the third-party layer arithmetic, drives them with
seeded inputs, asserts that ``oracle/qwen3tts_oracle.py`` reproduces them, and stores the
input/output vectors so that the GPU box (which has no ``/root/reference``) can check the HIP path
and the oracle against them.

    python oracle/make_golden.py            # regenerate + self-check

Pins produced
  sampler.npz    reference ``sample_logits`` / ``apply_repetition_penalty`` (sampling.py:10-66)
  stack.npz      transformers ``Qwen3OmniMoeTalkerCodePredictorModel`` (5.15 sibling of the qwen-tts
                 talker / predictor layers) driven with the predictor schedule (prefill 2, decode 1)
  decode.npz     reference ``fast_generate`` (generate.py:16-215) and ``fast_generate_streaming``
                 (streaming.py:19-188) run over duck-typed graph objects backed by the oracle's layer
                 math (``torch.cuda.synchronize`` patched to a no-op: the container has no GPU)
  codec.npz      transformers sibling vocoder blocks (CausalConvNet, CausalTransConvNet,
                 ConvNeXtBlock, SnakeBeta, DecoderBlock, Code2WavTransformerModel) + oracle full decode
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from fq3hip.config import tiny_test_config  # noqa: E402
from fq3hip.weights import synth_weights, synth_prompt  # noqa: E402
from oracle import qwen3tts_oracle as O  # noqa: E402


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
    sys.path.insert(0, REF)
    import importlib
    samp = importlib.import_module("faster_qwen3_tts.sampling")
    gen = importlib.import_module("faster_qwen3_tts.generate")
    stream = importlib.import_module("faster_qwen3_tts.streaming")
    torch.cuda.synchronize = lambda *a, **k: None   # no GPU here; the loops only use it as a fence
    return samp, gen, stream


def f32(t):
    return t.detach().to(torch.float32).cpu().numpy()


# --------------------------------------------------------------------------------------
def golden_sampler(samp):
    cases = []
    g = torch.Generator().manual_seed(7)
    cfgs = [
        dict(temperature=1.0, top_k=0, top_p=1.0, do_sample=False),
        dict(temperature=0.9, top_k=50, top_p=1.0, do_sample=True),
        dict(temperature=0.7, top_k=5, top_p=1.0, do_sample=True),
        dict(temperature=1.3, top_k=0, top_p=0.8, do_sample=True),
        dict(temperature=0.9, top_k=50, top_p=0.9, do_sample=True),
        dict(temperature=1.0, top_k=0, top_p=1.0, do_sample=True),
    ]
    V_list = [3072, 2048, 1104, 1280]
    n = 0
    n_same = 0
    for dt in (torch.float32, torch.bfloat16):
        for V in V_list:
            for ci, kw in enumerate(cfgs):
                for rep in range(3):
                    logits = (torch.randn(1, V, generator=g) * 3.0).to(dt)
                    if rep == 2:   # force ties at the k-th value and at the maximum
                        logits[0, 10:14] = logits.max()
                        logits[0, 100:110] = logits[0].float().topk(min(50, V))[0][-1].to(dt)
                    eos = V - 1024 + 102 if V > 1024 else V - 3
                    sm = O.build_suppress_mask(V, eos)
                    sup = [eos] if rep == 1 else None
                    seed = 1000 + n
                    torch.manual_seed(seed)
                    ref_tok = samp.sample_logits(logits, suppress_mask=sm, suppress_tokens=sup, **kw)
                    torch.manual_seed(seed)
                    noise = torch.empty(1, V, dtype=dt).exponential_(1)
                    my_tok = O.sample_logits(logits, suppress_mask=sm, suppress_tokens=sup, noise=noise,
                                             stable_top_p=False, **kw)
                    assert int(ref_tok) == int(my_tok), (dt, V, kw, rep, int(ref_tok), int(my_tok))
                    st_tok = O.sample_logits(logits, suppress_mask=sm, suppress_tokens=sup, noise=noise, **kw)
                    tie_free = kw["top_p"] >= 1.0 or (dt == torch.float32 and rep != 2)
                    if tie_free:     # stable == unstable whenever no tie group straddles the nucleus cut
                        assert int(st_tok) == int(ref_tok), (dt, V, kw, rep)
                    n_same += int(st_tok) == int(ref_tok)
                    cases.append(dict(logits=f32(logits[0]), noise=f32(noise[0]), V=V, eos=eos,
                                      sup_eos=int(sup is not None), bf16=int(dt == torch.bfloat16),
                                      token=int(st_tok), ref_token=int(ref_tok), **kw))
                    n += 1
    # repetition penalty (reference tests/test_sampling.py:10-21 KAT + random)
    pen = []
    for dt in (torch.float32, torch.bfloat16):
        for r in range(6):
            V = 3072
            logits = (torch.randn(1, 1, V, generator=g) * 3).to(dt)
            hist = torch.randint(0, V - 1024, (1 + 40 * r,), generator=g)
            p = 1.05 if r % 2 == 0 else 1.3
            ref = samp.apply_repetition_penalty(logits.clone(), hist, p)
            mine = O.apply_repetition_penalty(logits.clone(), hist, p)
            assert torch.equal(ref, mine)
            pen.append(dict(logits=f32(logits[0, 0]), hist=hist.numpy(), p=p, out=f32(ref[0, 0]),
                            bf16=int(dt == torch.bfloat16)))
    kat = torch.zeros(1, 1, 10); kat[..., 7] = 1.0; kat[..., 8] = -1.0
    out = O.apply_repetition_penalty(kat.clone(), torch.tensor([7, 8, 8, 1]), 1.1)
    assert abs(float(out[0, 0, 7]) - 1 / 1.1) < 1e-6 and abs(float(out[0, 0, 8]) + 1.1) < 1e-6
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), cases=np.array(cases, dtype=object),
                        penalty=np.array(pen, dtype=object))
    print(f"sampler.npz: {len(cases)} sampler cases ({n_same} identical under stable top-p tie order), "
          f"{len(pen)} penalty cases  [reference sampling.py == oracle]")


# --------------------------------------------------------------------------------------
def golden_stack():
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as M, configuration_qwen3_omni_moe as C
    from transformers import DynamicCache
    cfg = tiny_test_config()
    pc = cfg.predictor
    hc = C.Qwen3OmniMoeTalkerCodePredictorConfig(
        vocab_size=pc.vocab_size, hidden_size=pc.hidden_size, intermediate_size=pc.intermediate_size,
        num_hidden_layers=pc.num_hidden_layers, num_attention_heads=pc.num_attention_heads,
        num_key_value_heads=pc.num_key_value_heads, head_dim=pc.head_dim, rms_norm_eps=pc.rms_norm_eps,
        rope_parameters={"rope_type": "default", "rope_theta": pc.rope_theta}, num_code_groups=cfg.num_code_groups)
    hc._attn_implementation = "eager"
    out = {}
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        W = synth_weights(cfg, 0, dt)
        m = M.Qwen3OmniMoeTalkerCodePredictorModel(hc).eval().to(dt)
        pre = "talker.code_predictor.model."
        r = m.load_state_dict({k[len(pre):]: v for k, v in W.items() if k.startswith(pre)}, strict=False)
        assert not r.missing_keys and not r.unexpected_keys, r
        g = torch.Generator().manual_seed(11)
        xs = [torch.randn(1, 2, pc.hidden_size, generator=g).to(dt)] + \
             [torch.randn(1, 1, pc.hidden_size, generator=g).to(dt) for _ in range(5)]
        cache = DynamicCache(config=hc)
        oc = O.KVCache.empty(pc.num_hidden_layers, 17, pc.num_key_value_heads, pc.head_dim, dt)
        start = 0
        ys = []
        with torch.no_grad():
            for x in xs:
                ref = m(inputs_embeds=x, past_key_values=cache, use_cache=True).last_hidden_state[0]
                n = x.shape[1]
                mine = O.stack_forward(W, "talker.code_predictor.model", pc, x[0], start, oc,
                                       torch.arange(start, start + n).float())
                err = (ref.float() - mine.float()).abs().max().item()
                tol = 1e-5 if dt == torch.float32 else 0.08   # eager rounds scores/probs to bf16
                assert err <= tol, (tag, start, err)
                ys.append(ref)
                start += n
        out[f"x_{tag}"] = np.concatenate([f32(x[0]) for x in xs], 0)
        out[f"y_{tag}"] = np.concatenate([f32(y) for y in ys], 0)
    np.savez_compressed(os.path.join(OUT, "stack.npz"), **out)
    print("stack.npz: transformers sibling predictor model == oracle stack_forward (fp32 exact-ish, bf16 within eager tol)")


# --------------------------------------------------------------------------------------
class _DuckTalker:
    """Minimal duck type of the upstream talker that reference generate.py touches
    (cf. reference tests/test_sampling.py:52-93), backed by the oracle's layer math."""

    def __init__(self, orc: O.OracleTTS):
        self.o = orc
        self.rope_deltas = None
        W = orc.W
        self.codec_head = lambda h: torch.nn.functional.linear(h, W["talker.codec_head.weight"])
        emb = lambda name: (lambda ids: torch.nn.functional.embedding(ids, W[name]))
        self._embed = emb("talker.model.codec_embedding.weight")
        n = orc.cfg.num_code_groups - 1
        outer = self

        class CP:
            def get_input_embeddings(self_inner):
                return [emb(f"talker.code_predictor.model.codec_embedding.{i}.weight") for i in range(n)]
        self.code_predictor = CP()

    def get_input_embeddings(self):
        return self._embed

    def forward(self, inputs_embeds, attention_mask, **kw):
        logits, past_hidden, gen_step, L = self.o.prefill(inputs_embeds, attention_mask)
        return types.SimpleNamespace(past_key_values=L, past_hidden=past_hidden, generation_step=gen_step,
                                     logits=logits.view(1, 1, -1))


class _DuckTalkerGraph:
    def __init__(self, orc):
        self.o = orc
        self.max_seq_len = orc.max_seq_len

    def prefill_kv(self, past):      # the oracle prefill already wrote its own cache
        return past

    def set_generation_state(self, attention_mask, rope_deltas):
        pass

    def run(self, embeds, position):
        return self.o.talker_step(embeds, position)


class _DuckPredictorGraph:
    def __init__(self, orc):
        self.o = orc

    def run(self, pred_input):
        return self.o.predictor_loop(pred_input)


def golden_decode(gen, stream):
    cfg = tiny_test_config()
    out = {}
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        W = synth_weights(cfg, 0, dt)
        for case, (plen, tlen, maxnew, minnew, rp) in enumerate([(20, 8, 24, 0, 1.0), (33, 4, 16, 20, 1.0),
                                                                (12, 30, 20, 2, 1.05)]):
            tie, tam, tth, tpe, _ = synth_prompt(cfg, plen, tlen, 0, dtype=dt)
            orc = O.OracleTTS(cfg, W, max_seq_len=96)
            orc.pred_sampling = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
            conf = types.SimpleNamespace(codec_eos_token_id=cfg.codec_eos_token_id,
                                         num_code_groups=cfg.num_code_groups, vocab_size=cfg.talker.vocab_size)
            kw = dict(max_new_tokens=maxnew, min_new_tokens=minnew, temperature=1.0, top_k=0, top_p=1.0,
                      do_sample=False, repetition_penalty=rp)
            ref_codes, timing = gen.fast_generate(_DuckTalker(orc), tie, tam, tth, tpe, conf,
                                                  _DuckPredictorGraph(orc), _DuckTalkerGraph(orc), **kw)
            orc2 = O.OracleTTS(cfg, W, max_seq_len=96)
            orc2.pred_sampling = orc.pred_sampling
            sp = O.SamplingParams(**kw)
            mine = orc2.generate(tie, tam, tth, tpe, sp, record_margins=True)
            assert (ref_codes is None) == (mine is None)
            assert torch.equal(ref_codes, mine), (tag, case)
            # streaming == non-streaming (reference tests/test_e2e_parity.py:729-782)
            orc3 = O.OracleTTS(cfg, W, max_seq_len=96)
            orc3.pred_sampling = orc.pred_sampling
            chunks = list(stream.fast_generate_streaming(_DuckTalker(orc3), tie, tam, tth, tpe, conf,
                                                         _DuckPredictorGraph(orc3), _DuckTalkerGraph(orc3),
                                                         chunk_size=8, **kw))
            cat = torch.cat([c for c, _ in chunks], 0)
            assert torch.equal(cat, ref_codes)
            metas = [(t["chunk_index"], t["chunk_steps"], t["total_steps_so_far"], int(t["is_final"])) for _, t in chunks]
            out[f"codes_{tag}_{case}"] = ref_codes.numpy()
            out[f"margins_{tag}_{case}"] = np.array(orc2.margins, dtype=np.float32)
            out[f"pred_margins_{tag}_{case}"] = np.array(orc2.pred_margins, dtype=np.float32).reshape(-1, 15)
            out[f"chunks_{tag}_{case}"] = np.array(metas, dtype=np.int64)
            out[f"params_{tag}_{case}"] = np.array([plen, tlen, maxnew, minnew, rp], dtype=np.float64)
            print(f"  decode {tag} case {case}: {ref_codes.shape[0]} frames, min margin {min(orc2.margins):.4g}, "
                  f"chunks {metas}")
    np.savez_compressed(os.path.join(OUT, "decode.npz"), **out)
    print("decode.npz: reference fast_generate / fast_generate_streaming == oracle generate")


# --------------------------------------------------------------------------------------
def golden_codec():
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as M, configuration_qwen3_omni_moe as C
    cfg = tiny_test_config()
    c = cfg.codec
    W = synth_weights(cfg, 0, torch.float32, parts=("codec",))
    g = torch.Generator().manual_seed(5)
    L = c.latent_dim

    def cp(mod, mapping):
        sd = {k: W[v] for k, v in mapping.items()}
        r = mod.load_state_dict(sd, strict=True)
        return mod.eval()

    x = torch.randn(1, L, 13, generator=g)
    with torch.no_grad():
        # causal conv k7
        m = cp(M.Qwen3OmniMoeCausalConvNet(L, c.decoder_dim, 7), {"conv.weight": "decoder.decoder.0.conv.weight",
                                                                "conv.bias": "decoder.decoder.0.conv.bias"})
        e = (m(x)[0] - O.causal_conv1d(x[0], W["decoder.decoder.0.conv.weight"], W["decoder.decoder.0.conv.bias"])).abs().max()
        assert e < 1e-5, e
        # upsample transposed conv (k == s)
        m = cp(M.Qwen3OmniMoeCausalTransConvNet(L, L, 2, 2), {"conv.weight": "decoder.upsample.0.0.conv.weight",
                                                             "conv.bias": "decoder.upsample.0.0.conv.bias"})
        e = (m(x)[0] - O.causal_trans_conv1d(x[0], W["decoder.upsample.0.0.conv.weight"],
                                             W["decoder.upsample.0.0.conv.bias"], 2)).abs().max()
        assert e < 1e-5, e
        # ConvNeXt
        m = M.Qwen3OmniMoeConvNeXtBlock(L)
        pfx = "decoder.upsample.0.1"
        m = cp(m, {"dwconv.conv.weight": f"{pfx}.dwconv.conv.weight", "dwconv.conv.bias": f"{pfx}.dwconv.conv.bias",
                   "norm.weight": f"{pfx}.norm.weight", "norm.bias": f"{pfx}.norm.bias",
                   "pwconv1.weight": f"{pfx}.pwconv1.weight", "pwconv1.bias": f"{pfx}.pwconv1.bias",
                   "pwconv2.weight": f"{pfx}.pwconv2.weight", "pwconv2.bias": f"{pfx}.pwconv2.bias",
                   "gamma": f"{pfx}.gamma"})
        e = (m(x)[0] - O.convnext_block(x[0], W, pfx)).abs().max()
        assert e < 1e-4, e
        # decoder block 0 (SnakeBeta + transposed conv k=2r + 3 residual units)
        hc = C.Qwen3OmniMoeCode2WavConfig(
            codebook_size=c.codebook_size, hidden_size=c.hidden_size, num_attention_heads=c.num_attention_heads,
            num_key_value_heads=c.num_attention_heads, sliding_window=c.sliding_window,
            intermediate_size=c.intermediate_size, layer_scale_initial_scale=0.01, rms_norm_eps=c.rms_norm_eps,
            num_hidden_layers=c.num_hidden_layers, num_quantizers=c.num_quantizers,
            upsample_rates=list(c.upsample_rates), upsampling_ratios=list(c.upsampling_ratios),
            decoder_dim=c.decoder_dim, rope_parameters={"rope_type": "default", "rope_theta": c.rope_theta},
            head_dim=c.head_dim)
        hc._attn_implementation = "eager"
        blk = M.Qwen3OmniMoeCode2WavDecoderBlock(hc, 0)
        names = {}
        b = "decoder.decoder.1.block"
        names["block.0.alpha"] = f"{b}.0.alpha"; names["block.0.beta"] = f"{b}.0.beta"
        names["block.1.conv.weight"] = f"{b}.1.conv.weight"; names["block.1.conv.bias"] = f"{b}.1.conv.bias"
        for j in (2, 3, 4):
            for s in ("act1.alpha", "act1.beta", "conv1.conv.weight", "conv1.conv.bias", "act2.alpha", "act2.beta",
                      "conv2.conv.weight", "conv2.conv.bias"):
                names[f"block.{j}.{s}"] = f"{b}.{j}.{s}"
        blk = cp(blk, names)
        xd = torch.randn(1, c.decoder_dim, 9, generator=g)
        ref = blk(xd)[0]
        mine = O.decoder_block(xd[0], W, b, c.upsample_rates[0])
        assert ref.shape == mine.shape == (c.decoder_dim // 2, (9 - 1) * c.upsample_rates[0]), (ref.shape, mine.shape)
        e = (ref - mine).abs().max()
        assert e < 1e-4, e
        # transformer core (sliding window, layer scale)
        tm = M.Qwen3OmniMoeCode2WavTransformerModel(hc)
        t = "decoder.pre_transformer."
        sd = {k[len(t):]: v for k, v in W.items()
              if k.startswith(t) and not k[len(t):].startswith(("input_proj.", "output_proj."))}
        r = tm.load_state_dict(sd, strict=False)
        assert not r.unexpected_keys and not r.missing_keys, r
        tm.eval()
        T = 21
        xt = torch.randn(1, T, c.hidden_size, generator=g)
        ref = tm(inputs_embeds=xt).last_hidden_state[0]
        mine = O.codec_transformer_core(xt[0], W, c)
        e = (ref - mine).abs().max()
        assert e < 1e-4, e
    # full decode (oracle) stored as regression vector, fp32 and bf16
    out = {}
    codes = torch.randint(0, c.codebook_size, (11, c.num_quantizers), generator=g)
    out["codes"] = codes.numpy()
    out["wav_f32"] = f32(O.codec_decode(codes, W, c))
    Wb = synth_weights(cfg, 0, torch.bfloat16, parts=("codec",))
    out["wav_bf16"] = f32(O.codec_decode(codes, Wb, c))
    np.savez_compressed(os.path.join(OUT, "codec.npz"), **out)
    print(f"codec.npz: sibling vocoder blocks == oracle blocks; full decode {out['wav_f32'].shape[0]} samples for 11 frames, "
          f"|wav| max {np.abs(out['wav_f32']).max():.3f} rms {np.sqrt((out['wav_f32']**2).mean()):.3f}")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    # import the transformers siblings BEFORE the soundfile stub goes in (transformers probes it)
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as _M  # noqa: F401
    samp, gen, stream = _import_reference()
    torch.set_num_threads(4)
    golden_sampler(samp)
    golden_stack()
    golden_decode(gen, stream)
    golden_codec()
    print("all pins hold")
