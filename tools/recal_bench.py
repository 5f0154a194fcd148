#!/usr/bin/env python3
"""BN recalibration (batch_forward, train-mode BatchNorm with batch statistics) throughput against the eval forward on
the same batches: the step that precedes extraction for every subnet of a NAS sweep."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json  # noqa: E402
import torch  # noqa: E402

import sugar_amd  # noqa: E402
from sugar_amd import _lib, layout, synth  # noqa: E402

dev = torch.device("cuda", 0)
sup = sugar_amd.tdnn8m2g()
sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
sub = sup.clone(layout.BASE).set_precision("bf16")
model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev)
for B, L in ((128, 64000), (256, 48000), (32, 64000)):
    wav = torch.randn(B, L, device=dev)
    for mode in ("eval", "train"):
        model.train(mode == "train")
        for _ in range(3):
            model(wav, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            model(wav, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("%-5s %3d x %.0f s: %.3f ms per batch, %.0f utt/s" % (mode, B, L / 16000, dt * 1e3, B / dt), flush=True)
    model.eval()
