#!/usr/bin/env python3
"""batch_forward (BN recalibration) over 24 batches of 256 x 3 s: batches in pageable HOST memory (what a DataLoader
yields) against the same batches resident on the GPU."""
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
host = [(torch.randn(256, 48000),) for _ in range(24)]
resident = [(b[0].to(dev),) for b in host]
stats = {}
for name, loader in (("resident", resident), ("host", host), ("resident", resident), ("host", host)):
    sub = sup.clone(layout.BASE).set_precision("bf16")
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).eval()
    sugar_amd.batch_forward(loader[:2], model, device="cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sugar_amd.batch_forward(loader, model, device="cuda:0")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sd = {k: v.clone() for k, v in sub.state_dict().items() if "running" in k}
    same = "" if name not in stats else "  (statistics bit-equal to the first %s run: %s)" % (
        name, all(torch.equal(sd[k], stats[name][k]) for k in sd))
    stats.setdefault(name, sd)
    print("batch_forward, 24 x 256 x 3 s, batches %-8s: %.1f ms (%.2f ms per batch)%s" % (name, dt * 1e3, dt / 24 * 1e3, same),
          flush=True)
# what the loop cost before: a synchronous pageable copy per batch on the compute stream
sub = sup.clone(layout.BASE).set_precision("bf16")
model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).train()
for b in host[:2]:
    model(b[0].to(dev), None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in host:
    model(b[0].to(dev), None)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("the same 24 host batches with tensor.to(device) per batch: %.1f ms (%.2f ms per batch)" % (dt * 1e3, dt / 24 * 1e3))
print("host-fed statistics bit-equal to the resident run: %s" % all(torch.equal(stats["host"][k], stats["resident"][k])
                                                                     for k in stats["host"]))
