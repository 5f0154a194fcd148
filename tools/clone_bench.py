#!/usr/bin/env python3
"""NAS-sweep overhead (BASELINE config 5): time to clone a sampled subnet from the supernet and pack it for the GPU
(materialise + sugar_subnet_create), against the time its 4 874 utterances take to embed."""
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
path = sugar_amd.TDNNPath(sup.space(mode=["kernel", "depth", "width1", "width2"]))
wav = torch.randn(256, 48000, device=dev)
fb = sugar_amd.LogMelFbanks()
for i in range(6):
    cfg = path.sample() if i else layout.BASE
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sub = sup.clone(cfg).set_precision("bf16")
    t1 = time.perf_counter()
    model = sugar_amd.SpeakerModel(sub, transform=fb).to(dev).eval()
    model(wav[:2], normalize=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(2):
        model(wav, normalize=True)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    for _ in range(19):
        model(wav, normalize=True)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("cfg %s\n   clone %.1f ms, pack+first forward %.1f ms, 19 x 256 utterances %.1f ms" % (
        (cfg[0], list(cfg[1]), list(cfg[2]), cfg[3]), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3), flush=True)
