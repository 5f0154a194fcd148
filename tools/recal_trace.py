#!/usr/bin/env python3
"""One recalibration workload for a kernel trace: 20 train-mode forwards of Base on 256 x 3 s (rocprofv3 --kernel-trace)."""
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
model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev)
wav = torch.randn(256, 48000, device=dev)
model.train()
for _ in range(5):
    model(wav, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    model(wav, None)
torch.cuda.synchronize()
print("train-mode 256 x 3 s: %.3f ms per batch" % ((time.perf_counter() - t0) / 20 * 1e3))
