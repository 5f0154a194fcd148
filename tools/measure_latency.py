#!/usr/bin/env python3
"""Measure the MI355X latency table of the coarse tdnn8m2g search space (TDNNLatency.measure) and write it as yaml."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import layout  # noqa: E402
from sugar_amd.latency import TDNNLatency  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tdnn8m2g_latency_mi355x_80_300.yml"
if os.path.exists(out):
    os.remove(out)
t0 = time.time()
lat = TDNNLatency()
lat.measure(device="cuda:0", batch=64, frames=300, verbose=True)
lat.to_yaml(out)
n = sum(len(v) for v in lat.latency_table.values())
print("%d entries in %.1f s -> %s" % (n, time.time() - t0, out))
for name, cfg in (("base", layout.BASE), ("largest", layout.LARGEST), ("smallest", layout.SMALLEST)):
    print("query %s: %.4f ms per 64 x 300 frames" % (name, lat.query(cfg)))
