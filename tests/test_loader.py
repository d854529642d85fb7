"""CPU: the HF-checkpoint loader (fq3hip.weights.load_hf_checkpoint) on a tiny synthetic checkpoint written
in the recalled upstream layout (config.json + *.safetensors + speech_tokenizer/ sub-checkpoint with
EuclideanCodebook embedding_sum / cluster_usage buffers)."""
import json
import os

import pytest
import torch

from fq3hip.config import tiny_test_config
from fq3hip.weights import synth_weights, load_hf_checkpoint


def _write_checkpoint(root, cfg, W):
    from safetensors.torch import save_file
    t, p, c = cfg.talker, cfg.predictor, cfg.codec

    def stack(s):
        return dict(hidden_size=s.hidden_size, intermediate_size=s.intermediate_size, num_hidden_layers=s.num_hidden_layers,
                    num_attention_heads=s.num_attention_heads, num_key_value_heads=s.num_key_value_heads,
                    head_dim=s.head_dim, vocab_size=s.vocab_size, rms_norm_eps=s.rms_norm_eps, rope_theta=s.rope_theta)
    conf = dict(tts_pad_token_id=cfg.tts_pad_token_id, tts_bos_token_id=cfg.tts_bos_token_id,
                tts_eos_token_id=cfg.tts_eos_token_id, tts_model_type="base", tts_model_size="0b6",
                talker_config=dict(stack(t), num_code_groups=cfg.num_code_groups, codec_eos_token_id=cfg.codec_eos_token_id,
                                   codec_pad_id=cfg.codec_pad_id, codec_bos_id=cfg.codec_bos_id,
                                   codec_language_id=cfg.codec_language_id, text_vocab_size=cfg.text_vocab_size,
                                   text_hidden_size=cfg.text_hidden_size, code_predictor_config=stack(p)))
    os.makedirs(os.path.join(root, "speech_tokenizer"))
    json.dump(conf, open(os.path.join(root, "config.json"), "w"))
    main = {k: v.contiguous() for k, v in W.items() if not k.startswith("decoder.")}
    save_file(main, os.path.join(root, "model.safetensors"))
    dec = {}
    for k, v in W.items():
        if not k.startswith("decoder."):
            continue
        if k.endswith("._codebook.embedding"):      # upstream stores the running sums, not the embedding itself
            usage = torch.full((v.shape[0],), 2.0)
            dec[k + "_sum"] = (v.float() * 2.0).contiguous()
            dec[k[: -len("embedding")] + "cluster_usage"] = usage
        else:
            dec[k] = v.contiguous()
    save_file(dec, os.path.join(root, "speech_tokenizer", "model.safetensors"))
    dconf = dict(decoder_config=dict(codebook_size=c.codebook_size, codebook_dim=c.codebook_dim, rvq_dim=c.rvq_dim,
                                     latent_dim=c.latent_dim, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                                     num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                                     head_dim=c.head_dim, sliding_window=c.sliding_window, decoder_dim=c.decoder_dim,
                                     upsample_rates=list(c.upsample_rates), upsampling_ratios=list(c.upsampling_ratios)))
    json.dump(dconf, open(os.path.join(root, "speech_tokenizer", "config.json"), "w"))


def test_load_hf_checkpoint_roundtrip(tmp_path):
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("talker", "predictor", "codec", "text"))
    root = str(tmp_path / "ckpt")
    os.makedirs(root)
    _write_checkpoint(root, cfg, W)
    cfg2, W2 = load_hf_checkpoint(root, dtype=torch.float32)
    assert cfg2.talker.hidden_size == cfg.talker.hidden_size and cfg2.predictor.vocab_size == cfg.predictor.vocab_size
    assert cfg2.codec.decoder_dim == cfg.codec.decoder_dim and tuple(cfg2.codec.upsample_rates) == tuple(cfg.codec.upsample_rates)
    assert cfg2.codec_eos_token_id == cfg.codec_eos_token_id and cfg2.tts_pad_token_id == cfg.tts_pad_token_id
    for k, v in W.items():
        assert k in W2, k
        assert torch.allclose(W2[k].float(), v.float(), atol=1e-6), k


def test_load_hf_checkpoint_missing_tensor_is_loud(tmp_path):
    from safetensors.torch import save_file
    root = str(tmp_path / "bad")
    os.makedirs(root)
    json.dump({"talker_config": {}}, open(os.path.join(root, "config.json"), "w"))
    save_file({"talker.model.norm.weight": torch.ones(4)}, os.path.join(root, "model.safetensors"))
    with pytest.raises(KeyError, match="lacks tensors"):
        load_hf_checkpoint(root)


def test_from_pretrained_rejects_non_directories():
    from fq3hip.model import FasterQwen3TTS
    if not torch.cuda.is_available():
        with pytest.raises(ValueError, match="CUDA graphs require CUDA device"):
            FasterQwen3TTS.from_pretrained("Qwen/Qwen3-TTS-12Hz-0.6B-Base")
    else:
        with pytest.raises(FileNotFoundError):
            FasterQwen3TTS.from_pretrained("Qwen/Qwen3-TTS-12Hz-0.6B-Base")
