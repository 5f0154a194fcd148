#!/usr/bin/env python3
"""Throughput of the ragged path (BASELINE config 4: utterances of 2-20 s, no padding rows): utterances/s and audio
seconds/s per stage set, against the fixed 3 s batch.  Diagnostic; `SUGAR_GENERIC_RES2=1` forces the per-step Res2
path for an A/B in the same job."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sugar_amd  # noqa: E402
from sugar_amd import _lib, layout, synth  # noqa: E402
import json  # noqa: E402

dev = torch.device("cuda", 0)
sup = sugar_amd.tdnn8m2g()
sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
sub = sup.clone(layout.BASE).set_precision("bf16")
model = sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks()).to(dev).eval()

rng = np.random.default_rng(0)
cases = {
    "fixed 3 s x 256": [48000] * 256,
    "2-6 s x 192": list(rng.integers(32000, 96000, 192)),
    "2-20 s x 96": list(rng.integers(32000, 320000, 96)),
    "6.5-20 s x 64": list(rng.integers(104000, 320000, 64)),
    "60 s x 16": [960000] * 16,
    "3 s x 1": [48000],
    "8 s x 1": [128000],
    "3 s x 8": [48000] * 8,
    "145 s x 1": [2320000],
}
for name, lens in cases.items():
    wavs = [torch.randn(int(n), device=dev) for n in lens]
    for _ in range(3):
        model.embed_ragged(wavs, normalize=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        model.embed_ragged(wavs, normalize=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    secs = sum(lens) / 16000.0
    print("%-18s %8.3f ms  %9.0f utt/s  %9.0f audio-s/s  (%.0f s of audio per batch)" % (name, dt * 1e3, len(lens) / dt,
                                                                                         secs / dt, secs), flush=True)
    _lib.profile_enable((1 << _lib.STAGE_COUNT) - 1)
    for _ in range(5):
        model.embed_ragged(wavs, normalize=True)
    torch.cuda.synchronize()
    print("    stages ms: " + json.dumps({k: round(ms / 5, 3) for k, (c, ms) in _lib.profile_read().items()}), flush=True)
    _lib.profile_enable(0)
