#!/usr/bin/env python3
"""A/B: one 256-utterance pass vs the same utterances in sub-batches (working set vs the 256 MB Infinity Cache)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sugar_amd  # noqa: E402
from sugar_amd import layout, synth  # noqa: E402

dev = torch.device("cuda", 0)
sup = sugar_amd.tdnn8m2g()
sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
sub = sup.clone(layout.BASE).set_precision("bf16")
model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).eval()
wav = torch.randn(256, 48000, device=dev)
for rep in range(2):
    for chunk in (256, 128, 96, 64, 32):
        parts = [wav[i:i + chunk].contiguous() for i in range(0, 256, chunk)]
        for _ in range(3):
            for p in parts:
                model(p, normalize=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            for p in parts:
                model(p, normalize=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("chunk %3d: %.3f ms per 256 utterances  (%.0f emb/s)" % (chunk, dt * 1e3, 256 / dt), flush=True)
