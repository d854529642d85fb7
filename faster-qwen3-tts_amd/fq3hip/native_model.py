"""Native stand-in for the upstream ``qwen_tts.Qwen3TTSModel`` object graph the reference wrapper holds.

The reference keeps ``base_model`` (a ``qwen_tts`` object, third party, not installable offline) and
reaches into it for the talker, its embeddings, the prompt helpers and the speech tokenizer
(SURVEY.md Appendix D lists every access).  This module provides the same attribute surface over a
plain weight table so that ``FasterQwen3TTS`` and the decode loops run without ``qwen_tts``:

* hot path (talker / predictor / sampler / codec decoder): HIP, through ``Fq3Engine`` / ``HipSpeechTokenizer``;
* prompt building (text embedding + ``text_projection`` MLP, codec prefix embeddings, ICL reference
  code embedding sum): HIP (``fq3_text_project`` / ``fq3_prompt_rows``, driven by ``fq3hip/prompt.py``); the
  module-style accessors below (``get_text_embeddings``, ``text_projection``, ``generate_icl_prompt``) keep the
  attribute surface the reference's generic code path reaches for;
* reference-audio analysis (speaker encoder, speech-tokenizer *encoder*): HIP (``fq3hip/refenc.py`` over
  ``csrc/fq3_refenc.hip``) when the weight table carries ``encoder.*`` / ``speaker_encoder.*`` tensors; otherwise callers
  pass a precomputed ``voice_clone_prompt`` (the reference supports exactly that, model.py:320-411) or a voice cache.

``generate_icl_prompt`` / the chat templates restate upstream behaviour from memory of
``qwen_tts/core/models/modeling_qwen3_tts.py`` [recalled, unverifiable offline].
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, Dict, List, Optional

import torch
import torch.nn.functional as F

from .codec import HipSpeechTokenizer
from .config import TTSConfig
from .engine import Fq3Engine

Weights = Dict[str, torch.Tensor]


@dataclass
class VoiceClonePromptItem:
    """Same fields as upstream's prompt item (reference model.py:336-352 reads these attributes)."""
    ref_code: Optional[torch.Tensor]
    ref_spk_embedding: torch.Tensor
    x_vector_only_mode: bool
    icl_mode: bool
    ref_text: Optional[str] = None


class ByteTokenizer:
    """Deterministic stand-in tokenizer for synthetic-weight models: 3 header ids, one id per UTF-8
    byte, 5 trailer ids -- the slicing contract of the prompt builder (model.py:686, 701-702, 718)."""

    def __init__(self, vocab: int):
        self.vocab = vocab

    def __call__(self, text: str) -> List[int]:
        body = [16 + (b % (self.vocab - 32)) for b in text.encode("utf-8")]
        return [1, 2, 3] + body + [4, 5, 1, 2, 3]


class _Embedding:
    def __init__(self, weight: torch.Tensor):
        self.weight = weight

    def __call__(self, ids: torch.Tensor) -> torch.Tensor:
        return F.embedding(ids.to(self.weight.device), self.weight)


class NativeTalker:
    def __init__(self, cfg: TTSConfig, engine: Fq3Engine, text_weights: Optional[Weights], share: Optional["NativeTalker"] = None):
        self.cfg, self.engine = cfg, engine
        self.device = engine.device
        self.rope_deltas = None
        self.config = SimpleNamespace(
            hidden_size=cfg.talker.hidden_size, num_hidden_layers=cfg.talker.num_hidden_layers,
            vocab_size=cfg.talker.vocab_size, num_code_groups=cfg.num_code_groups,
            codec_eos_token_id=cfg.codec_eos_token_id, codec_pad_id=cfg.codec_pad_id, codec_bos_id=cfg.codec_bos_id,
            codec_think_id=cfg.codec_think_id, codec_nothink_id=cfg.codec_nothink_id,
            codec_think_bos_id=cfg.codec_think_bos_id, codec_think_eos_id=cfg.codec_think_eos_id,
            codec_language_id=dict(cfg.codec_language_id), spk_id=dict(cfg.spk_id),
            spk_is_dialect=dict(cfg.spk_is_dialect))
        self._codec_emb = _Embedding(engine.codec_embedding)
        self._pred_embs = [_Embedding(w) for w in engine.predictor_embeddings]
        self.code_predictor = SimpleNamespace(get_input_embeddings=lambda: list(self._pred_embs))
        self._tw = None
        if text_weights is not None:
            dt, dev = engine.dtype, engine.device
            if share is not None and share._tw is not None and share.engine.dtype == dt and share.device == dev:
                self._tw = share._tw                     # one device copy of the 600 MB text table per GPU
            else:
                self._tw = {k: v.to(device=dev, dtype=dt) for k, v in text_weights.items()}
            self._text_emb = _Embedding(self._tw["talker.model.text_embedding.weight"])
            w = self._tw
            engine.bind_prompt_weights(w["talker.model.text_embedding.weight"], w["talker.text_projection.linear_fc1.weight"],
                                       w["talker.text_projection.linear_fc1.bias"], w["talker.text_projection.linear_fc2.weight"],
                                       w["talker.text_projection.linear_fc2.bias"])
        self.hip_prompt_ready = self._tw is not None

    def get_input_embeddings(self):
        return self._codec_emb

    def get_text_embeddings(self):
        if self._tw is None:
            raise RuntimeError("this model was built without the text embedding / projection weights")
        return self._text_emb

    def text_projection(self, x: torch.Tensor) -> torch.Tensor:
        w = self._tw
        h = F.linear(x, w["talker.text_projection.linear_fc1.weight"], w["talker.text_projection.linear_fc1.bias"])
        return F.linear(F.silu(h), w["talker.text_projection.linear_fc2.weight"], w["talker.text_projection.linear_fc2.bias"])

    def codec_head(self, hidden: torch.Tensor) -> torch.Tensor:
        return self.engine.codec_head(hidden.reshape(-1).contiguous()).view(1, -1)


class NativeInner:
    """``base_model.model`` in the reference's terms."""

    def __init__(self, cfg: TTSConfig, talker: NativeTalker, speech_tokenizer: Optional[HipSpeechTokenizer]):
        self.talker = talker
        self.speech_tokenizer = speech_tokenizer
        self.tts_model_type = cfg.tts_model_type
        self.tts_model_size = cfg.tts_model_size
        self.config = SimpleNamespace(talker_config=talker.config, tts_bos_token_id=cfg.tts_bos_token_id,
                                      tts_eos_token_id=cfg.tts_eos_token_id, tts_pad_token_id=cfg.tts_pad_token_id)
        self._cfg = cfg

    def generate_speaker_prompt(self, voice_clone_prompt: Dict[str, Any]) -> List[torch.Tensor]:
        t = self.talker
        return [torch.as_tensor(e).to(device=t.device, dtype=t.engine.dtype) for e in voice_clone_prompt["ref_spk_embedding"]]

    def generate_icl_prompt(self, text_id, ref_id, ref_code, tts_pad_embed, tts_eos_embed, non_streaming_mode):
        """[recalled upstream behaviour] text stream = ref text + target text + eos; codec stream =
        codec_bos + per-frame sum of the 16 codebook embeddings of the reference codes."""
        t, cfg = self.talker, self._cfg
        dev = t.device
        text_embed = t.text_projection(t.get_text_embeddings()(torch.cat([ref_id, text_id], dim=-1).to(dev)))
        text_embed = torch.cat([text_embed, tts_eos_embed], dim=1)
        ref_code = ref_code.to(dev)
        embs = [t.get_input_embeddings()(ref_code[:, :1])]
        pe = t.code_predictor.get_input_embeddings()
        for i in range(1, cfg.num_code_groups):
            embs.append(pe[i - 1](ref_code[:, i:i + 1]))
        codec_embed = torch.cat(embs, dim=1).sum(1).unsqueeze(0)
        bos = t.get_input_embeddings()(torch.tensor([[cfg.codec_bos_id]], device=dev))
        codec_embed = torch.cat([bos, codec_embed], dim=1)
        tl, cl = text_embed.shape[1], codec_embed.shape[1]
        if non_streaming_mode:
            pad = t.get_input_embeddings()(torch.tensor([[cfg.codec_pad_id] * tl], device=dev))
            return torch.cat([text_embed + pad, codec_embed + tts_pad_embed], dim=1), tts_pad_embed
        if tl > cl:
            return text_embed[:, :cl] + codec_embed, text_embed[:, cl:]
        text_embed = torch.cat([text_embed] + [tts_pad_embed] * (cl - tl), dim=1)
        return text_embed + codec_embed, tts_pad_embed


class NativeQwen3TTS:
    """``base_model`` in the reference's terms: text helpers + ``.model``."""

    def __init__(self, cfg: TTSConfig, weights: Weights, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                 max_seq_len: int = 2048, tokenizer=None, codec_max_frames: int = 1024, max_frames: int = 2048,
                 share: Optional["NativeQwen3TTS"] = None, codec_precision: Optional[str] = None):
        """``codec_precision``: arithmetic of the 12 Hz codec decoder -- ``None`` / ``"model"`` = the model dtype (bf16 for a bf16
        checkpoint, what the reference's Torch path runs), ``"fp32"`` = the checkpoint's weights widened exactly to fp32, fp32
        activations and fp32 matrix-core products (v_mfma_f32_16x16x4_f32): the waveform of the fp32 oracle to ~1e-6 RMS at ~4x
        the decode time (the decode loop keeps the model dtype either way)."""
        if codec_precision not in (None, "model", "bf16", "fp32", "bf16x2"):
            raise ValueError("codec_precision must be None, 'model', 'bf16', 'fp32' or 'bf16x2'")
        codec_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "bf16x2": torch.float32}.get(codec_precision, dtype)
        codec_prec = codec_precision if codec_precision in ("bf16", "fp32", "bf16x2") else None
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = dtype
        self.engine = Fq3Engine(cfg, weights, device=device, dtype=dtype, max_seq_len=max_seq_len, max_frames=max_frames,
                                share=share.engine if share is not None else None)
        text = {k: v for k, v in weights.items() if k.startswith(("talker.model.text_embedding", "talker.text_projection"))}
        talker = NativeTalker(cfg, self.engine, text if text else None, share=share.model.talker if share is not None else None)
        tok = None
        if any(k.startswith("decoder.") for k in weights):
            tok = HipSpeechTokenizer(cfg.codec, weights, device=device, dtype=codec_dtype, max_frames=codec_max_frames,
                                     share=share.model.speech_tokenizer if share is not None else None, precision=codec_prec)
        self.model = NativeInner(cfg, talker, tok)
        self.tokenizer = tokenizer or ByteTokenizer(cfg.text_vocab_size)
        # reference-audio analysers (first call per reference clip): one instance per GPU, shared by every lane
        self.ref_analyzer = None
        if share is not None and getattr(share, "ref_analyzer", None) is not None and share.device == self.device:
            self.ref_analyzer = share.ref_analyzer
        else:
            ref = {k: v for k, v in weights.items() if k.startswith(("encoder.", "speaker_encoder."))}
            if ref:
                from .refenc import HipRefAudioAnalyzer
                self.ref_analyzer = HipRefAudioAnalyzer(cfg.ref_audio, ref, device=device)
        if tok is not None and self.ref_analyzer is not None and self.ref_analyzer.has_encoder:
            tok.attach_encoder(self.ref_analyzer)

    # ---- text helpers (upstream chat templates, [recalled]) ----------------------------------------
    @staticmethod
    def _build_assistant_text(text: str) -> str:
        return f"<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n"

    @staticmethod
    def _build_ref_text(text: str) -> str:
        return f"<|im_start|>assistant\n{text}<|im_end|>\n"

    @staticmethod
    def _build_instruct_text(instruct: str) -> str:
        return f"<|im_start|>user\n{instruct}<|im_end|>\n"

    def _tokenize_texts(self, texts: List[str]) -> List[torch.Tensor]:
        out = []
        for t in texts:
            if isinstance(self.tokenizer, ByteTokenizer):
                # strip the template: the byte tokenizer supplies its own header / trailer ids
                body = t.split("\n", 1)[1].rsplit("<|im_end|>", 1)[0] if "<|im_start|>" in t else t
                ids = self.tokenizer(body)
                if not t.endswith("assistant\n"):       # ref / instruct turns end after "<|im_end|>\n": 2 trailer ids
                    ids = ids[:-3]
            else:
                ids = self.tokenizer(t)["input_ids"] if not callable(getattr(self.tokenizer, "encode", None)) \
                    else self.tokenizer.encode(t)
            tt = torch.tensor([ids], dtype=torch.long, device=self.device)
            # the list just uploaded, noted for the prompt builder (prompt.host_ids): valid while the tensor is not written to
            tt.fq3_host_ids = (list(ids), None if tt.is_inference() else tt._version)
            out.append(tt)
        return out

    # ---- validation -------------------------------------------------------------------------------------
    def get_supported_speakers(self):
        return sorted(self.cfg.spk_id.keys())

    def get_supported_languages(self):
        return ["auto"] + sorted(self.cfg.codec_language_id.keys())

    def _validate_languages(self, languages):
        ok = set(self.get_supported_languages())
        for l in languages:
            if l is not None and l.lower() not in ok:
                raise ValueError(f"Unsupported language {l!r}; supported: {sorted(ok)}")

    def _validate_speakers(self, speakers):
        ok = set(self.get_supported_speakers())
        for s in speakers:
            if s is None or s.lower() not in ok:
                raise ValueError(f"Unsupported speaker {s!r}; supported: {sorted(ok)}")

    # ---- voice-clone prompt handling -----------------------------------------------------------------------
    def _reference_wave(self, ref_audio) -> "np.ndarray":
        """``ref_audio``: a WAV path, ``(waveform, sample_rate)`` (what the reference passes, model.py:443-447) or a
        waveform already at the model rate -> mono float32 at 24 kHz."""
        import numpy as np
        from .audio_io import load_audio, resample
        rate = self.cfg.ref_audio.sample_rate
        if isinstance(ref_audio, (str, os.PathLike)):
            audio, sr = load_audio(str(ref_audio))          # same loader as model._load_ref_audio_with_silence
        elif isinstance(ref_audio, (tuple, list)) and len(ref_audio) == 2 and not np.isscalar(ref_audio[0]):
            audio, sr = ref_audio
        else:
            audio, sr = ref_audio, rate
        audio = np.asarray(audio.detach().cpu().numpy() if hasattr(audio, "detach") else audio, dtype=np.float32)
        if audio.ndim > 1:
            audio = audio.mean(axis=-1) if audio.shape[-1] <= 8 else audio.mean(axis=0)
        return resample(audio, int(sr), rate)

    def create_voice_clone_prompt(self, ref_audio=None, ref_text: str = "", x_vector_only_mode: bool = False):
        """Upstream ``Qwen3TTSModel.create_voice_clone_prompt`` [recalled] over the HIP analysers: the x-vector always,
        the reference codes in ICL mode (``x_vector_only_mode=False``, which needs ``ref_text``)."""
        an = self.ref_analyzer
        if an is None or not an.has_speaker:
            raise NotImplementedError(
                "This weight table has no speaker_encoder.* / encoder.* tensors, so reference audio cannot be analysed here. "
                "Pass voice_clone_prompt=dict(ref_spk_embedding=[...], ref_code=[...], x_vector_only_mode=[...], icl_mode=[...]) "
                "or serve ref_audio from a voice-reference cache (FasterQwen3TTS.set_voice_ref_cache).")
        if ref_audio is None:
            raise ValueError("ref_audio is required")
        icl = not x_vector_only_mode
        if icl and not ref_text:
            raise ValueError("ref_text is required when x_vector_only_mode=False (ICL mode).")
        if icl and not an.has_encoder:
            raise NotImplementedError("ICL mode needs the speech tokenizer's encoder.* tensors; use x_vector_only_mode=True")
        wav = self._reference_wave(ref_audio)
        spk = an.speaker_embedding(wav).to(self.dtype)
        codes = an.encode(wav) if icl else None
        return [VoiceClonePromptItem(ref_code=codes, ref_spk_embedding=spk, x_vector_only_mode=bool(x_vector_only_mode),
                                     icl_mode=icl, ref_text=ref_text if icl else None)]

    @staticmethod
    def _prompt_items_to_voice_clone_prompt(items: List[Any]) -> Dict[str, Any]:
        return dict(ref_code=[it.ref_code for it in items], ref_spk_embedding=[it.ref_spk_embedding for it in items],
                    x_vector_only_mode=[bool(it.x_vector_only_mode) for it in items],
                    icl_mode=[bool(it.icl_mode) for it in items])
