"""Prompt builder on the HIP path: the same rows the reference's ``_build_talker_inputs_local``
(``faster_qwen3_tts/model.py:583-805``) and upstream ``generate_icl_prompt`` (called at ``model.py:699-712``) assemble,
computed with three launches of arithmetic instead of ~40 small ATen ops:

* every text token of the prompt (specials, instruct turn, role header, reference text, target text) goes through
  ``text_projection(text_embedding(.))`` in ONE batched MLP on the matrix cores (``fq3_text_project``);
* one kernel assembles all prompt rows and all trailing-text rows from a row program (``fq3_prompt_rows``).

What stays on the host is control flow only: which token / which codec id / which reference frame goes to which row --
the layout table of SURVEY.md Appendix B, written out as integers.  Batch size 1 (the public API never passes more,
``model.py:494-495``); anything else takes the generic tensor path in ``model.py``.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

NONE, TOKEN, SPEAKER, REF_FRAME = 0, 1, 2, 3


class _TextPool:
    """Collects token ids; returns the row index each one will have in the batched text_projection output."""

    def __init__(self):
        self.ids: List[int] = []

    def add(self, ids) -> List[int]:
        ids = [int(x) for x in ids]
        base = len(self.ids)
        self.ids += ids
        return list(range(base, base + len(ids)))


def host_ids(t: torch.Tensor) -> List[int]:
    """The token ids of ``t`` on the host.  The tokenising step (native_model._tokenize_texts) notes the list it uploaded on the tensor
    -- with the tensor's version counter, like the padding note of the prefill -- so a prompt build costs no device-to-host copy (three
    stream waits per request before round 6; in a running scheduler each waited for the prefills queued ahead).  A tensor from anywhere
    else, or written to since, is read back."""
    note = getattr(t, "fq3_host_ids", None)
    if isinstance(note, tuple) and len(note) == 2 and (note[1] is None or (not t.is_inference() and note[1] == t._version)):
        return [int(x) for x in note[0]]
    return [int(x) for x in t.reshape(-1).tolist()]


def build_talker_inputs_hip(m, input_id: torch.Tensor, ref_id: Optional[torch.Tensor], voice_clone_prompt, index: int,
                            language: str, speaker: Optional[str], non_streaming_mode: bool,
                            instruct_id: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """One utterance -> (talker_input_embeds [1, L, H], attention_mask [1, L], trailing_text_hiddens [1, T', H],
    tts_pad_embed [1, 1, H]) on the device, like the reference's function returns for B = 1."""
    tk, tc, mc = m.talker, m.config.talker_config, m.config
    eng = tk.engine
    iid = host_ids(input_id)
    pool = _TextPool()
    i_bos, i_eos, i_pad = pool.add([mc.tts_bos_token_id, mc.tts_eos_token_id, mc.tts_pad_token_id])
    rows: List[Tuple[int, int, int]] = []          # (text_row, kind, arg)
    trail: List[Tuple[int, int, int]] = []

    # optional instruct turn: text only (model.py:601-606)
    if instruct_id is not None:
        rows += [(t, NONE, 0) for t in pool.add(host_ids(instruct_id))]

    # speaker embedding of the codec prefix (model.py:615-631)
    spk_vec, spk_kind = None, None
    if voice_clone_prompt is not None:
        if voice_clone_prompt["x_vector_only_mode"][index] or voice_clone_prompt["icl_mode"][index]:
            spk_vec = m.generate_speaker_prompt(voice_clone_prompt)[index]
            spk_kind = (SPEAKER, 0)
    elif speaker not in ("", None):
        if speaker.lower() not in tc.spk_id:
            raise NotImplementedError(f"Speaker {speaker} not implemented")
        spk_kind = (TOKEN, int(tc.spk_id[speaker.lower()]))

    # language id / dialect override (model.py:633-650)
    assert language is not None
    if language.lower() == "auto":
        lang_id = None
    else:
        if language.lower() not in tc.codec_language_id:
            raise NotImplementedError(f"Language {language} not implemented")
        lang_id = tc.codec_language_id[language.lower()]
    if language.lower() in ("chinese", "auto") and speaker not in ("", None) and tc.spk_is_dialect.get(speaker.lower()):
        lang_id = tc.codec_language_id[tc.spk_is_dialect[speaker.lower()]]

    # role header (3 text rows), then the codec prefix against [tts_pad ... tts_pad, tts_bos] (model.py:657-697)
    rows += [(t, NONE, 0) for t in pool.add(iid[:3])]
    prefix = ([tc.codec_nothink_id, tc.codec_think_bos_id, tc.codec_think_eos_id] if lang_id is None else
              [tc.codec_think_id, tc.codec_think_bos_id, lang_id, tc.codec_think_eos_id])
    cod = [(TOKEN, int(c)) for c in prefix]
    if spk_kind is not None:
        cod.append(spk_kind)
    cod.append((TOKEN, int(tc.codec_pad_id)))          # the trailing codec_bos is held back
    for j, (kind, arg) in enumerate(cod):
        rows.append((i_bos if j == len(cod) - 1 else i_pad, kind, arg))

    ref_codes = None
    icl = (voice_clone_prompt is not None and voice_clone_prompt.get("ref_code", None) is not None
           and voice_clone_prompt["icl_mode"][index])
    if icl:
        # upstream generate_icl_prompt [recalled]: text stream = ref text + target text + eos; codec stream = codec_bos +
        # one 16-codebook embedding sum per reference frame
        ref_codes = torch.as_tensor(voice_clone_prompt["ref_code"][index])
        rid = host_ids(ref_id)
        text = pool.add(rid[3:-2] + iid[3:-5]) + [i_eos]
        codec = [(TOKEN, int(tc.codec_bos_id))] + [(REF_FRAME, f) for f in range(int(ref_codes.shape[0]))]
        tl, cl = len(text), len(codec)
        if non_streaming_mode:
            rows += [(t, TOKEN, int(tc.codec_pad_id)) for t in text]
            rows += [(i_pad, k, a) for k, a in codec]
            trail = [(i_pad, NONE, 0)]
        elif tl > cl:
            rows += [(text[j], codec[j][0], codec[j][1]) for j in range(cl)]
            trail = [(t, NONE, 0) for t in text[cl:]]
        else:
            rows += [(text[j] if j < tl else i_pad, codec[j][0], codec[j][1]) for j in range(cl)]
            trail = [(i_pad, NONE, 0)]
    elif non_streaming_mode:
        body = pool.add(iid[3:-5]) + [i_eos]
        rows += [(t, TOKEN, int(tc.codec_pad_id)) for t in body]                     # model.py:724-747
        rows.append((i_pad, TOKEN, int(tc.codec_bos_id)))
        trail = [(i_pad, NONE, 0)]
    else:
        first = pool.add(iid[3:4])
        rows.append((first[0] if first else -1, TOKEN, int(tc.codec_bos_id)))         # model.py:714-723
        trail = [(t, NONE, 0) for t in pool.add(iid[4:-5])] + [(i_eos, NONE, 0)]      # model.py:758-766

    dev = eng.device
    text_rows = eng.text_project(torch.tensor(pool.ids, dtype=torch.long, device=dev))
    prog = torch.tensor(rows + trail + [(i_pad, NONE, 0)], dtype=torch.int32, device=dev)
    out = eng.prompt_rows(text_rows, prog, ref_codes=ref_codes, spk_embed=spk_vec)
    L, Tt = len(rows), len(trail)
    tie = out[:L].unsqueeze(0)
    tth = out[L:L + Tt].unsqueeze(0)
    tpe = out[L + Tt:].unsqueeze(0)
    tam = torch.ones(1, L, dtype=torch.long, device=dev)
    # host-side note for the prefill: a single prompt is never padded -- saves a device reduction + host wait per request.  The note
    # carries the tensor's version counter: an in-place edit of the mask afterwards voids it (generate._n_pad_of counts again)
    # (a tensor created under torch.inference_mode() has no version counter: its note is taken as it is)
    tam.fq3_n_pad = (0, None if tam.is_inference() else tam._version)
    return tie, tam, tth, tpe
