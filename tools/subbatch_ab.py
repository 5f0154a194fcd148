#!/usr/bin/env python3
"""A/B: one 256-utterance pass vs the same utterances as sub-batches, sequentially or on concurrent HIP streams
(memory-bound kernels of one sub-batch under the MFMA-bound kernels of another)."""
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


def make():
    sub = sup.clone(layout.BASE).set_precision("bf16")
    return sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).eval()


wav = torch.randn(256, 48000, device=dev)
for rep in range(2):
    for nstreams, chunk in ((1, 256), (2, 128), (2, 256), (3, 256), (4, 256)):
        models = [make() for _ in range(nstreams)]           # one workspace per stream
        streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
        total = 256 if chunk != 256 else 256 * nstreams
        src = torch.cat([wav] * (total // 256))
        parts = [src[i:i + chunk].contiguous() for i in range(0, total, chunk)]

        def run():
            for i, p in enumerate(parts):
                with torch.cuda.stream(streams[i % nstreams]):
                    models[i % nstreams](p, normalize=True)

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("streams %d chunk %3d: %.3f ms per %d utterances  (%.0f emb/s)" % (nstreams, chunk, dt * 1e3, total,
                                                                               total / dt), flush=True)
