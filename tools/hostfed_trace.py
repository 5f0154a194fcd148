#!/usr/bin/env python3
"""One host-fed extraction for a timeline (rocprofv3 --kernel-trace --memory-copy-trace): do the PCIe copies overlap the kernels?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sugar_amd  # noqa: E402
from sugar_amd import layout, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
DEV = "cuda:0"
sup = sugar_amd.tdnn8m2g()
sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
sub = sup.clone(layout.BASE).set_precision("bf16")
model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()
lens = synth.ragged_lengths(N, lo=32000, hi=320000, seed=4).tolist()
g = torch.Generator().manual_seed(4)
wavs = [torch.randn(n, generator=g) for n in lens]


class Utts:
    lengths = lens

    def __len__(self):
        return N

    def __getitem__(self, i):
        return {"file": "u%d" % i, "input": wavs[i].unsqueeze(0)}


for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sugar_amd.extract_vectors(model, Utts(), norm=True, device=DEV, batch_size=256, n_jobs=8)
    torch.cuda.synchronize()
    print("rep %d: %.3f s" % (rep, time.perf_counter() - t0), flush=True)
    print("MARK %d %d" % (rep, time.time_ns()), flush=True)
