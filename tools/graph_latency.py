"""Eager vs HIP-graph replay latency of one embedding forward at small batches (host wall clock around a synchronised
call, median of 200).  python tools/graph_latency.py > profiles/r02_graph_latency.log"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sugar_amd
from sugar_amd import layout, synth
from sugar_amd.graphs import GraphedEmbedder

DEV = "cuda:0"


def main():
    supernet = sugar_amd.tdnn8m2g(in_feats=80, out_embeds=192)
    supernet.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0), strict=False)
    for precision in ("bf16", "fp32"):
        sub = supernet.clone(layout.BASE).set_precision(precision)
        supernet.config = None
        model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()
        for B in (1, 4, 16, 64):
            wav = synth.waveforms(B, 48000, seed=B).to(DEV)
            g = GraphedEmbedder(model, wav, normalize=True)

            def timed(fn, n=200):
                ts = []
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                for _ in range(n):
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                return statistics.median(ts) * 1e3
            with torch.no_grad():
                e = timed(lambda: model(wav, normalize=True))
                r = timed(lambda: g(wav))
            same = torch.equal(g(wav), model(wav, normalize=True))
            print("Base %s  %2d x 3 s: eager %.3f ms, graph replay %.3f ms per call (%.2fx), outputs identical: %s"
                  % (precision, B, e, r, e / r, same), flush=True)


if __name__ == "__main__":
    main()
