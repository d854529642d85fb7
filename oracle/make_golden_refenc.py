#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generates tests/golden/refenc.npz: outputs of the transformers modules the reference-audio analysers
are modelled on (MimiModel.encode, ECAPA_TimeDelayNet.forward), run HERE on CPU with the seeded synthetic weights of
fq3hip.weights.synth_ref_audio_weights at the REAL shapes, for the waveforms tests/test_gpu_refenc.py rebuilds from their
seeds.  The GPU test then compares the HIP kernels with these module outputs directly (not through the restatement);
oracle margins are stored only to tell a near-tie arg-min from a kernel error.

    python oracle/make_golden_refenc.py        # needs transformers (present in the image); ~1 min on CPU
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))

from fq3hip.config import RefAudioConfig                                    # noqa: E402
from fq3hip.weights import synth_ref_audio_weights                          # noqa: E402
from oracle import refenc_oracle as RO                                      # noqa: E402
from tests.test_refenc_oracle_pins import _ecapa, _mimi, make_wave          # noqa: E402

ENC = dict(weights_seed=1, n=24000 * 3 + 2500, wave_seed=7)                 # = tests/test_gpu_refenc.py::test_tokenizer_encoder_real_shapes
SPK = dict(weights_seed=2, n=24000 * 4 + 123, wave_seed=11)                 # = ...::test_speaker_encoder[False]


def main():
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    rc = RefAudioConfig()
    out = {}
    with torch.no_grad():
        W = synth_ref_audio_weights(rc, ENC["weights_seed"])
        x = make_wave(ENC["n"], seed=ENC["wave_seed"])
        hf = _mimi(rc, W).encode(x.reshape(1, 1, -1), num_quantizers=rc.num_quantizers).audio_codes[0].transpose(0, 1)
        codes, margins = RO.tokenizer_encode(W, rc, x, return_all=True)[:2]
        print(f"encoder: {tuple(hf.shape)} codes, restatement agrees with MimiModel on {int((codes == hf).sum())}/{hf.numel()} ids, "
              f"smallest oracle margin {float(margins.min()):.3e}")
        out.update(enc_meta=np.array([ENC["weights_seed"], ENC["n"], ENC["wave_seed"]]), enc_codes_mimi=hf.numpy().astype(np.int16),
                   enc_margins=margins.numpy().astype(np.float32))
        W = synth_ref_audio_weights(rc, SPK["weights_seed"])
        x = make_wave(SPK["n"], seed=SPK["wave_seed"])
        emb, mel = RO.speaker_embedding(W, rc, x)
        ref = _ecapa(rc, W)(mel[None])[0]
        print(f"speaker: ECAPA_TimeDelayNet vs restatement max |diff| {float((ref - emb).abs().max()):.3e} on scale {float(ref.abs().max()):.3f}")
        out.update(spk_meta=np.array([SPK["weights_seed"], SPK["n"], SPK["wave_seed"]]), spk_xvector_ecapa=ref.numpy().astype(np.float32),
                   spk_mel_first_frames=mel[:8].numpy().astype(np.float32))
    path = os.path.join(ROOT, "tests", "golden", "refenc.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
