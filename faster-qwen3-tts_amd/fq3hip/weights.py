"""Weight tables: seeded synthetic weights at real shapes, and the HF-checkpoint loader.

Tensor names follow the upstream checkpoint layout the reference reaches through
``base_model.model.talker`` (reference accesses at
``faster_qwen3_tts/predictor_graph.py:54-58``, ``generate.py:100-101``,
``model.py:605``).  Synthetic initialisation is the protocol of SURVEY.md section 8(d):
``torch.manual_seed``-style CPU generator, Linear/conv ~ N(0, 1/fan_in), embeddings and
codebooks ~ N(0, 1); norm gains / SnakeBeta / layer-scale are perturbed away from their
trivial values so a kernel that ignores them fails parity.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, Optional

import torch

from .config import TTSConfig, StackConfig, CodecConfig, RefAudioConfig, from_hf_config

Weights = Dict[str, torch.Tensor]


class _Gen:
    def __init__(self, seed: int):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(seed)

    def normal(self, *shape, std=1.0):
        return torch.randn(*shape, generator=self.g, dtype=torch.float32) * std

    def linear(self, out_f, in_f, fan_in=None):
        fan_in = fan_in or in_f
        return self.normal(out_f, in_f, std=fan_in ** -0.5)

    def gain(self, n, spread=0.1):
        return 1.0 + self.normal(n, std=spread)


def _stack(w: Weights, g: _Gen, prefix: str, c: StackConfig):
    H, I = c.hidden_size, c.intermediate_size
    for i in range(c.num_hidden_layers):
        p = f"{prefix}.layers.{i}"
        w[f"{p}.input_layernorm.weight"] = g.gain(H)
        w[f"{p}.self_attn.q_proj.weight"] = g.linear(c.q_dim, H)
        w[f"{p}.self_attn.k_proj.weight"] = g.linear(c.kv_dim, H)
        w[f"{p}.self_attn.v_proj.weight"] = g.linear(c.kv_dim, H)
        w[f"{p}.self_attn.o_proj.weight"] = g.linear(H, c.q_dim)
        w[f"{p}.self_attn.q_norm.weight"] = g.gain(c.head_dim)
        w[f"{p}.self_attn.k_norm.weight"] = g.gain(c.head_dim)
        w[f"{p}.post_attention_layernorm.weight"] = g.gain(H)
        w[f"{p}.mlp.gate_proj.weight"] = g.linear(I, H)
        w[f"{p}.mlp.up_proj.weight"] = g.linear(I, H)
        w[f"{p}.mlp.down_proj.weight"] = g.linear(H, I)
    w[f"{prefix}.norm.weight"] = g.gain(H)


def _codec(w: Weights, g: _Gen, c: CodecConfig, normalized: bool = False):
    """``normalized``: damp the residual branches (conv2 x 0.3) so that activations stay O(1) through the 12 residual units
    -- like a trained vocoder, and unlike the default variance-doubling trunk whose O(100) SnakeBeta phases make the
    waveform chaotic in bf16 -- and size the output conv for a PCM std of ~0.1."""
    p = "decoder"
    rg = 0.3 if normalized else 1.0
    nq, ns = c.num_quantizers, c.num_semantic_quantizers
    for name, n in (("rvq_first", ns), ("rvq_rest", nq - ns)):
        for j in range(n):
            w[f"{p}.quantizer.{name}.vq.layers.{j}._codebook.embedding"] = g.normal(c.codebook_size, c.rvq_dim)
        w[f"{p}.quantizer.{name}.output_proj.weight"] = g.linear(c.codebook_dim, c.rvq_dim).unsqueeze(-1)
    L, Hc = c.latent_dim, c.hidden_size
    w[f"{p}.pre_conv.conv.weight"] = g.normal(L, c.codebook_dim, 3, std=(c.codebook_dim * 3) ** -0.5)
    w[f"{p}.pre_conv.conv.bias"] = g.normal(L, std=0.02)
    t = f"{p}.pre_transformer"
    w[f"{t}.input_proj.weight"] = g.linear(Hc, L)
    w[f"{t}.input_proj.bias"] = g.normal(Hc, std=0.02)
    w[f"{t}.output_proj.weight"] = g.linear(L, Hc)
    w[f"{t}.output_proj.bias"] = g.normal(L, std=0.02)
    qd = c.num_attention_heads * c.head_dim
    for i in range(c.num_hidden_layers):
        q = f"{t}.layers.{i}"
        w[f"{q}.input_layernorm.weight"] = g.gain(Hc)
        w[f"{q}.self_attn.q_proj.weight"] = g.linear(qd, Hc)
        w[f"{q}.self_attn.k_proj.weight"] = g.linear(qd, Hc)
        w[f"{q}.self_attn.v_proj.weight"] = g.linear(qd, Hc)
        w[f"{q}.self_attn.o_proj.weight"] = g.linear(Hc, qd)
        w[f"{q}.self_attn_layer_scale.scale"] = 0.5 + g.normal(Hc, std=0.05)
        w[f"{q}.post_attention_layernorm.weight"] = g.gain(Hc)
        w[f"{q}.mlp.gate_proj.weight"] = g.linear(c.intermediate_size, Hc)
        w[f"{q}.mlp.up_proj.weight"] = g.linear(c.intermediate_size, Hc)
        w[f"{q}.mlp.down_proj.weight"] = g.linear(Hc, c.intermediate_size)
        w[f"{q}.mlp_layer_scale.scale"] = 0.5 + g.normal(Hc, std=0.05)
    w[f"{t}.norm.weight"] = g.gain(Hc)
    for i, f in enumerate(c.upsampling_ratios):
        u = f"{p}.upsample.{i}"
        # ConvTranspose1d weight layout: [in, out, k]
        w[f"{u}.0.conv.weight"] = g.normal(L, L, f, std=L ** -0.5)
        w[f"{u}.0.conv.bias"] = g.normal(L, std=0.02)
        w[f"{u}.1.dwconv.conv.weight"] = g.normal(L, 1, 7, std=7 ** -0.5)
        w[f"{u}.1.dwconv.conv.bias"] = g.normal(L, std=0.02)
        w[f"{u}.1.norm.weight"] = g.gain(L)
        w[f"{u}.1.norm.bias"] = g.normal(L, std=0.02)
        w[f"{u}.1.pwconv1.weight"] = g.linear(4 * L, L)
        w[f"{u}.1.pwconv1.bias"] = g.normal(4 * L, std=0.02)
        w[f"{u}.1.pwconv2.weight"] = g.linear(L, 4 * L)
        w[f"{u}.1.pwconv2.bias"] = g.normal(L, std=0.02)
        w[f"{u}.1.gamma"] = 0.5 + g.normal(L, std=0.05)
    d = f"{p}.decoder"
    D = c.decoder_dim
    w[f"{d}.0.conv.weight"] = g.normal(D, L, 7, std=(L * 7) ** -0.5)
    w[f"{d}.0.conv.bias"] = g.normal(D, std=0.02)
    for i, r in enumerate(c.upsample_rates):
        cin, cout = D // 2 ** i, D // 2 ** (i + 1)
        b = f"{d}.{i + 1}.block"
        w[f"{b}.0.alpha"] = g.normal(cin, std=0.1)
        w[f"{b}.0.beta"] = g.normal(cin, std=0.1)
        w[f"{b}.1.conv.weight"] = g.normal(cin, cout, 2 * r, std=(2 * cin) ** -0.5)
        w[f"{b}.1.conv.bias"] = g.normal(cout, std=0.02)
        for j in range(3):
            u = f"{b}.{j + 2}"
            w[f"{u}.act1.alpha"] = g.normal(cout, std=0.1)
            w[f"{u}.act1.beta"] = g.normal(cout, std=0.1)
            w[f"{u}.conv1.conv.weight"] = g.normal(cout, cout, 7, std=(cout * 7) ** -0.5)
            w[f"{u}.conv1.conv.bias"] = g.normal(cout, std=0.02)
            w[f"{u}.act2.alpha"] = g.normal(cout, std=0.1)
            w[f"{u}.act2.beta"] = g.normal(cout, std=0.1)
            w[f"{u}.conv2.conv.weight"] = g.normal(cout, cout, 1, std=cout ** -0.5) * rg
            w[f"{u}.conv2.conv.bias"] = g.normal(cout, std=0.02) * rg
    n = len(c.upsample_rates)
    cl = D // 2 ** n
    w[f"{d}.{n + 1}.alpha"] = g.normal(cl, std=0.1)
    w[f"{d}.{n + 1}.beta"] = g.normal(cl, std=0.1)
    # the synthetic trunk has O(50) activations at its end; scale the output conv so PCM stays inside
    # [-1, 1] (std ~0.15) and the final clamp does not hide kernel errors from the parity tests
    w[f"{d}.{n + 2}.conv.weight"] = g.normal(1, cl, 7, std=(cl * 7) ** -0.5 * (0.06 if normalized else 0.0033))
    w[f"{d}.{n + 2}.conv.bias"] = g.normal(1, std=0.01)


def synth_ref_audio_weights(rc: RefAudioConfig, seed: int = 0, device: str = "cpu") -> Weights:
    """Seeded fp32 weights for the reference-audio analysers, under the checkpoint names: the speech tokenizer's encoder
    as ``encoder.<MimiModel state-dict name>`` and the speaker encoder as ``speaker_encoder.<ECAPA state-dict name>``
    [recalled prefixes].  Conv/linear ~ N(0, 1/fan_in); layer scales 0.5 (0.01 would hide the transformer); codebooks
    as (embed_sum, cluster_usage) pairs like ``MimiEuclideanCodebook`` stores them."""
    w: Weights = {}
    g = _Gen(seed * 1000 + 5)
    conv = lambda co, ci, k: g.normal(co, ci, k, std=(ci * k) ** -0.5)
    bias = lambda n: g.normal(n, std=0.05)
    E = "encoder.encoder.layers"
    C = rc.num_filters
    w[f"{E}.0.conv.weight"] = conv(C, 1, rc.kernel_size) * 4.0          # waveforms are O(0.1): lift them to O(1)
    w[f"{E}.0.conv.bias"] = bias(C)
    li = 1
    for r in rc.ratios:
        for j in range(rc.num_residual_layers):
            w[f"{E}.{li}.block.1.conv.weight"] = conv(C // rc.compress, C, rc.residual_kernel_size)
            w[f"{E}.{li}.block.1.conv.bias"] = bias(C // rc.compress)
            w[f"{E}.{li}.block.3.conv.weight"] = conv(C, C // rc.compress, 1)
            w[f"{E}.{li}.block.3.conv.bias"] = bias(C)
            li += 1
        li += 1
        w[f"{E}.{li}.conv.weight"] = conv(2 * C, C, 2 * r)
        w[f"{E}.{li}.conv.bias"] = bias(2 * C)
        li += 1
        C *= 2
    li += 1
    H = rc.hidden_size
    w[f"{E}.{li}.conv.weight"] = conv(H, C, rc.last_kernel_size)
    w[f"{E}.{li}.conv.bias"] = bias(H)
    QD = rc.num_attention_heads * rc.head_dim
    for l in range(rc.num_hidden_layers):
        p = f"encoder.encoder_transformer.layers.{l}"
        for n in ("q_proj", "k_proj", "v_proj"):
            w[f"{p}.self_attn.{n}.weight"] = g.linear(QD, H)
        w[f"{p}.self_attn.o_proj.weight"] = g.linear(H, QD)
        w[f"{p}.mlp.fc1.weight"] = g.linear(rc.intermediate_size, H)
        w[f"{p}.mlp.fc2.weight"] = g.linear(H, rc.intermediate_size)
        for n in ("input_layernorm", "post_attention_layernorm"):
            w[f"{p}.{n}.weight"] = g.gain(H)
            w[f"{p}.{n}.bias"] = bias(H)
        w[f"{p}.self_attn_layer_scale.scale"] = 0.5 + g.normal(H, std=0.05)
        w[f"{p}.mlp_layer_scale.scale"] = 0.5 + g.normal(H, std=0.05)
    w["encoder.downsample.conv.weight"] = conv(H, H, 4)
    D, K = rc.codebook_dim, rc.codebook_size
    for name, n in (("semantic", rc.num_semantic_quantizers), ("acoustic", rc.num_quantizers - rc.num_semantic_quantizers)):
        q = f"encoder.quantizer.{name}_residual_vector_quantizer"
        w[f"{q}.input_proj.weight"] = conv(D, H, 1)
        for i in range(n):
            usage = 0.5 + 1.5 * torch.rand(K, generator=g.g)
            # residual scales shrink level by level in a trained RVQ; keep every level's codebook comparable to its input
            w[f"{q}.layers.{i}.codebook.embed_sum"] = g.normal(K, D, std=0.8 ** i) * usage[:, None]
            w[f"{q}.layers.{i}.codebook.cluster_usage"] = usage
    S = "speaker_encoder"
    ch, ks = rc.enc_channels, rc.enc_kernel_sizes
    w[f"{S}.blocks.0.conv.weight"] = conv(ch[0], rc.mel_dim, ks[0]) * 0.25      # log-mels are O(5)
    w[f"{S}.blocks.0.conv.bias"] = bias(ch[0])
    for b in range(1, len(ch) - 1):
        B = f"{S}.blocks.{b}"
        cs = ch[b] // rc.enc_res2net_scale
        for t in ("tdnn1", "tdnn2"):
            w[f"{B}.{t}.conv.weight"] = conv(ch[b], ch[b], 1)
            w[f"{B}.{t}.conv.bias"] = bias(ch[b])
        for i in range(rc.enc_res2net_scale - 1):
            w[f"{B}.res2net_block.blocks.{i}.conv.weight"] = conv(cs, cs, ks[b])
            w[f"{B}.res2net_block.blocks.{i}.conv.bias"] = bias(cs)
        w[f"{B}.se_block.conv1.weight"] = conv(rc.enc_se_channels, ch[b], 1)
        w[f"{B}.se_block.conv1.bias"] = bias(rc.enc_se_channels)
        w[f"{B}.se_block.conv2.weight"] = conv(ch[b], rc.enc_se_channels, 1)
        w[f"{B}.se_block.conv2.bias"] = bias(ch[b])
    cm = ch[-1]
    w[f"{S}.mfa.conv.weight"] = conv(cm, cm, 1)
    w[f"{S}.mfa.conv.bias"] = bias(cm)
    w[f"{S}.asp.tdnn.conv.weight"] = conv(rc.enc_attention_channels, 3 * cm, 1)
    w[f"{S}.asp.tdnn.conv.bias"] = bias(rc.enc_attention_channels)
    w[f"{S}.asp.conv.weight"] = conv(cm, rc.enc_attention_channels, 1) * 4.0     # a peaked attention, so the softmax matters
    w[f"{S}.asp.conv.bias"] = bias(cm)
    w[f"{S}.fc.weight"] = conv(rc.enc_dim, 2 * cm, 1)
    w[f"{S}.fc.bias"] = bias(rc.enc_dim)
    return {k: v.to(device) for k, v in w.items()}


def synth_weights(cfg: TTSConfig, seed: int = 0, dtype: torch.dtype = torch.bfloat16,
                  device: str = "cpu", parts: Iterable[str] = ("talker", "predictor", "codec"),
                  codec_normalized: bool = False) -> Weights:
    """Seeded synthetic weights.  Each part has its own generator stream so that
    ``parts`` does not change the values of the parts that are generated."""
    w: Weights = {}
    t, pc = cfg.talker, cfg.predictor
    parts = tuple(parts)
    if "talker" in parts:
        g = _Gen(seed * 1000 + 1)
        w["talker.model.codec_embedding.weight"] = g.normal(t.vocab_size, t.hidden_size)
        _stack(w, g, "talker.model", t)
        w["talker.codec_head.weight"] = g.linear(t.vocab_size, t.hidden_size)
    if "predictor" in parts:
        g = _Gen(seed * 1000 + 2)
        pre = "talker.code_predictor"
        if cfg.predictor_has_projection:
            w[f"{pre}.small_to_mtp_projection.weight"] = g.linear(pc.hidden_size, t.hidden_size)
            w[f"{pre}.small_to_mtp_projection.bias"] = g.normal(pc.hidden_size, std=0.02)
        _stack(w, g, f"{pre}.model", pc)
        for j in range(cfg.num_code_groups - 1):
            w[f"{pre}.model.codec_embedding.{j}.weight"] = g.normal(pc.vocab_size, t.hidden_size)
            w[f"{pre}.lm_head.{j}.weight"] = g.linear(pc.vocab_size, pc.hidden_size)
    if "text" in parts:
        g = _Gen(seed * 1000 + 3)
        th = cfg.text_hidden_size
        w["talker.model.text_embedding.weight"] = g.normal(cfg.text_vocab_size, th)
        w["talker.text_projection.linear_fc1.weight"] = g.linear(th, th)
        w["talker.text_projection.linear_fc1.bias"] = g.normal(th, std=0.02)
        w["talker.text_projection.linear_fc2.weight"] = g.linear(t.hidden_size, th)
        w["talker.text_projection.linear_fc2.bias"] = g.normal(t.hidden_size, std=0.02)
    if "codec" in parts:
        _codec(w, _Gen(seed * 1000 + 4), cfg.codec, normalized=codec_normalized)
    return {k: v.to(dtype=dtype).to(device) for k, v in w.items()}


def synth_prompt(cfg: TTSConfig, prompt_len: int = 200, trailing_len: int = 32, ref_frames: int = 0,
                 dtype: torch.dtype = torch.bfloat16, device: str = "cpu", seed: int = 1234):
    """Synthetic prompt of SURVEY.md section 8(d): ``talker_input_embeds[1,L,H]`` ~ N(0,1)*0.02
    (seed), all-ones mask, ``trailing_text_hiddens[1,T',H]`` (seed+1), ``tts_pad_embed`` (seed+2),
    optional ICL ``ref_codes[T_ref,16]`` (seed+3)."""
    H = cfg.talker.hidden_size
    g = lambda s: _Gen(s)
    tie = (g(seed).normal(1, prompt_len, H) * 0.02).to(dtype).to(device)
    tam = torch.ones(1, prompt_len, dtype=torch.long, device=device)
    tth = (g(seed + 1).normal(1, trailing_len, H) * 0.02).to(dtype).to(device)
    tpe = (g(seed + 2).normal(1, 1, H) * 0.02).to(dtype).to(device)
    ref_codes = None
    if ref_frames > 0:
        gg = torch.Generator(device="cpu")
        gg.manual_seed(seed + 3)
        ref_codes = torch.randint(0, cfg.codec.codebook_size, (ref_frames, cfg.num_code_groups),
                                  generator=gg, dtype=torch.long).to(device)
    return tie, tam, tth, tpe, ref_codes


# --------------------------------------------------------------------------------------
# Real checkpoints
# --------------------------------------------------------------------------------------
def load_hf_checkpoint(path: str, dtype: torch.dtype = torch.bfloat16, device: str = "cpu"):
    """Load a local Qwen3-TTS checkpoint directory (``config.json`` + ``*.safetensors``, and the
    ``speech_tokenizer/`` sub-checkpoint) into ``(TTSConfig, Weights)``.

    This replaces ``Qwen3TTSModel.from_pretrained`` (``faster_qwen3_tts/model.py:192-197``) for the
    tensors the hot path needs.  No checkpoint exists in the build container, so the key mapping
    below is the recalled upstream layout and is validated only structurally:
    every tensor the kernels bind must be found, otherwise a ``KeyError`` names what is missing.
    """
    from safetensors.torch import load_file  # local import: optional dependency of this one entry

    with open(os.path.join(path, "config.json")) as f:
        cfg_dict = json.load(f)
    tok_dir = os.path.join(path, "speech_tokenizer")
    if os.path.isdir(tok_dir) and os.path.exists(os.path.join(tok_dir, "config.json")):
        with open(os.path.join(tok_dir, "config.json")) as f:
            tok_cfg = json.load(f)
        cfg_dict["decoder_config"] = tok_cfg.get("decoder_config")
        cfg_dict["encoder_config"] = tok_cfg.get("encoder_config")
        if tok_cfg.get("encoder_valid_num_quantizers"):
            cfg_dict["encoder_valid_num_quantizers"] = tok_cfg["encoder_valid_num_quantizers"]
    cfg = from_hf_config(cfg_dict)
    w: Weights = {}

    def ingest(d, strip=""):
        for fn in sorted(os.listdir(d)):
            if fn.endswith(".safetensors"):
                for k, v in load_file(os.path.join(d, fn)).items():
                    if strip and k.startswith(strip):
                        k = k[len(strip):]
                    # the EuclideanCodebook statistics stay fp32 until the division below (upstream divides in fp32)
                    keep32 = k.endswith("._codebook.embedding_sum") or k.endswith("._codebook.cluster_usage")
                    # the reference-audio analysers run in fp32 from the stored values (fq3hip/refenc.py)
                    if k.startswith(("encoder.", "speaker_encoder.")):
                        w[k] = v.to(device)
                        continue
                    w[k] = v.to(dtype=(torch.float32 if keep32 else dtype) if v.is_floating_point() else v.dtype).to(device)

    ingest(path)
    if os.path.isdir(tok_dir):
        ingest(tok_dir)
    # EuclideanCodebook stores (embedding_sum, cluster_usage); materialise the embedding.
    for k in [k for k in w if k.endswith("._codebook.embedding_sum")]:
        base = k[: -len("embedding_sum")]
        usage = w[base + "cluster_usage"].float().clamp(min=1e-5)
        w[base + "embedding"] = (w[k].float() / usage[:, None]).to(dtype)
        del w[k], w[base + "cluster_usage"]
    required = ["talker.model.codec_embedding.weight", "talker.codec_head.weight", "talker.model.norm.weight"]
    missing = [k for k in required if k not in w]
    if missing:
        raise KeyError(f"checkpoint at {path} lacks tensors required by the decode path: {missing}")
    return cfg, w
