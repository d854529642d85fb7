#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: the two reference-audio analysers on a 10.5 s clip, 5 times each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-qwen3-tts_amd"))
import numpy as np, torch
import bench
from fq3hip.config import qwen3_tts_0p6b
from fq3hip.refenc import HipRefAudioAnalyzer
from fq3hip.weights import synth_ref_audio_weights

rc = qwen3_tts_0p6b().ref_audio
an = HipRefAudioAnalyzer(rc, synth_ref_audio_weights(rc, 0))
x = torch.from_numpy(np.concatenate([bench.reference_wave(10.0), np.zeros(12000, np.float32)])).cuda()
for _ in range(5):
    an.encode(x); an.speaker_embedding(x)
torch.cuda.synchronize()
