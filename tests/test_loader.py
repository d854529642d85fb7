"""CPU: the HF-checkpoint loader (fq3hip.weights.load_hf_checkpoint) on a tiny synthetic checkpoint written
in the recalled upstream layout (config.json + *.safetensors + speech_tokenizer/ sub-checkpoint with
EuclideanCodebook embedding_sum / cluster_usage buffers)."""
import json
import os

import pytest
import torch

from fq3hip.config import tiny_test_config
from fq3hip.weights import synth_weights, synth_ref_audio_weights, load_hf_checkpoint


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
    ra = cfg.ref_audio
    if any(k.startswith("speaker_encoder.") for k in W):
        conf["speaker_encoder_config"] = dict(mel_dim=ra.mel_dim, enc_dim=ra.enc_dim, enc_channels=list(ra.enc_channels),
                                              enc_kernel_sizes=list(ra.enc_kernel_sizes), enc_dilations=list(ra.enc_dilations),
                                              enc_attention_channels=ra.enc_attention_channels, enc_res2net_scale=ra.enc_res2net_scale,
                                              enc_se_channels=ra.enc_se_channels, n_fft=ra.n_fft, hop_size=ra.hop_size)
    os.makedirs(os.path.join(root, "speech_tokenizer"))
    json.dump(conf, open(os.path.join(root, "config.json"), "w"))
    main = {k: v.contiguous() for k, v in W.items() if not k.startswith(("decoder.", "encoder."))}
    save_file(main, os.path.join(root, "model.safetensors"))
    dec = {}
    for k, v in W.items():
        if k.startswith("encoder."):               # the tokenizer's encoder half lives in the same sub-checkpoint
            dec[k] = v.contiguous()
            continue
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
    if any(k.startswith("encoder.") for k in W):
        dconf["encoder_valid_num_quantizers"] = ra.num_quantizers
        dconf["encoder_config"] = dict(num_filters=ra.num_filters, upsampling_ratios=list(reversed(ra.ratios)), kernel_size=ra.kernel_size,
                                       last_kernel_size=ra.last_kernel_size, residual_kernel_size=ra.residual_kernel_size,
                                       num_residual_layers=ra.num_residual_layers, dilation_growth_rate=ra.dilation_growth_rate,
                                       compress=ra.compress, hidden_size=ra.hidden_size, num_hidden_layers=ra.num_hidden_layers,
                                       num_attention_heads=ra.num_attention_heads, head_dim=ra.head_dim,
                                       intermediate_size=ra.intermediate_size, sliding_window=ra.sliding_window, norm_eps=ra.norm_eps,
                                       num_quantizers=32, num_semantic_quantizers=ra.num_semantic_quantizers,
                                       codebook_size=ra.codebook_size, codebook_dim=ra.codebook_dim,
                                       max_position_embeddings=ra.max_positions)
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


def test_load_hf_checkpoint_keeps_the_reference_audio_analysers(tmp_path):
    """speaker_encoder.* (main checkpoint) and encoder.* (speech_tokenizer/) arrive under their names, at their stored
    precision even when the model dtype is bf16, and their configs populate cfg.ref_audio."""
    from dataclasses import asdict
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, torch.float32, parts=("talker", "predictor", "codec", "text"))
    R = synth_ref_audio_weights(cfg.ref_audio, 4)
    root = str(tmp_path / "ckpt")
    os.makedirs(root)
    _write_checkpoint(root, cfg, {**W, **R})
    cfg2, W2 = load_hf_checkpoint(root, dtype=torch.bfloat16)
    assert asdict(cfg2.ref_audio) == asdict(cfg.ref_audio)
    for k, v in R.items():
        assert k in W2 and W2[k].dtype == torch.float32 and torch.equal(W2[k], v), k
    assert W2["talker.codec_head.weight"].dtype == torch.bfloat16
    # every tensor the packer binds is present (the packer indexes by name and raises KeyError otherwise)
    from fq3hip.refenc import pack_ref_audio_weights
    packed = pack_ref_audio_weights(W2, cfg2.ref_audio)
    assert "encoder.quantizer.codebook.15.embed_sum" in packed and "speaker_encoder.asp.tdnn.conv.weight_ms" in packed


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
