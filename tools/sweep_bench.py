#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: N sampled subnets, each embedding 4 874 synthetic 3 s utterances (resident on the GPU)
and scoring 37 720 synthetic trials; evaluate_subnets with and without the host-thread prefetch of the next subnet."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sugar_amd  # noqa: E402
from sugar_amd import layout, synth  # noqa: E402

N_SUB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N_UTT, N_TRIALS = 4874, 37720
dev = torch.device("cuda", 0)
sup = sugar_amd.tdnn8m2g()
sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
path = sugar_amd.TDNNPath(sup.space(mode=["kernel", "depth", "width1", "width2"]))
configs = [path.sample() for _ in range(N_SUB)]
wav = torch.randn(N_UTT, 48000, device=dev)


class Utts:
    def __len__(self):
        return N_UTT

    def __getitem__(self, i):
        return {"file": "utt%04d.wav" % i, "input": wav[i:i + 1]}


class Trials(list):
    utterances = None


class Loader:
    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size, self.num_workers = dataset, batch_size, 0


lab, a, b = synth.trial_pairs(N_UTT, N_TRIALS, seed=1)
trials = Trials({"label": int(lab[i]), "enroll": "utt%04d.wav" % a[i], "test": "utt%04d.wav" % b[i]} for i in range(N_TRIALS))
trials.utterances = Utts()
loader = Loader(trials, 256)
sugar_amd.evaluate_subnets(sup, configs[:2], loader, device="cuda:0")     # warm-up
for prefetch in (False, True, False, True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = sugar_amd.evaluate_subnets(sup, configs, loader, device="cuda:0", prefetch=prefetch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("prefetch %-5s: %d subnets x %d utterances + %d trials in %.3f s  (%.1f ms per subnet, %.0f emb/s)" % (
        prefetch, N_SUB, N_UTT, N_TRIALS, dt, dt / N_SUB * 1e3, N_SUB * N_UTT / dt), flush=True)
if os.environ.get("SWEEP_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    sugar_amd.evaluate_subnets(sup, configs[:4], loader, device="cuda:0", prefetch=False)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
