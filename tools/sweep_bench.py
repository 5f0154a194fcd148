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
if os.environ.get("SWEEP_LANES"):                 # A/B: HIP streams that consecutive batches of one extraction alternate over
    import sugar_amd.vectors
    sugar_amd.vectors.N_LANES = int(os.environ["SWEEP_LANES"])
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
loader = Loader(trials, int(os.environ.get("SWEEP_BATCH", "256")))      # utterances per packed call (A/B)
recal = None
if os.environ.get("SWEEP_RECAL"):      # the full NAS loop: BN recalibration over 24 x 256 x 3 s utterances before every evaluation
    rw = torch.randn(24, 256, 48000, device=dev)
    recal = [(rw[i], None) for i in range(24)]
sugar_amd.evaluate_subnets(sup, configs[:2], loader, recal_loader=recal, device="cuda:0")     # warm-up
for prefetch in (False, True, False, True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tm = {}
    res = sugar_amd.evaluate_subnets(sup, configs, loader, recal_loader=recal, device="cuda:0", prefetch=prefetch, timings=tm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("   launching thread, ms per subnet: " + ", ".join("%s %.1f" % (k, v / N_SUB * 1e3) for k, v in tm.items()), flush=True)
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
if os.environ.get("SWEEP_GPU"):
    # where a subnet's time goes: GPU time of the extraction alone (events around extract_vectors on a packed list), the
    # host time of clone + pack, and of scoring + metrics
    from sugar_amd.sweep import _prepare, _TrialSnapshot, _LoaderView
    from sugar_amd.vectors import PackedUtterances, extract_vectors
    packed = PackedUtterances(trials.utterances, dev, batch_size=256)
    view = _LoaderView(loader, _TrialSnapshot(trials))
    t_gpu = t_prep = t_val = 0.0
    for cfg in configs:
        t0 = time.perf_counter()
        sub = _prepare(sup, cfg, "bf16", dev)
        t_prep += time.perf_counter() - t0
        model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).eval()
        extract_vectors(model, packed)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        vec = extract_vectors(model, packed)
        e1.record()
        torch.cuda.synchronize()
        t_gpu += e0.elapsed_time(e1) * 1e-3
        t0 = time.perf_counter()
        sugar_amd.veri_validate(view, model, device="cuda:0", vectors=vec)
        t_val += time.perf_counter() - t0
        sub.release()
    print("per subnet: extraction on the GPU %.1f ms, clone + pack on the host %.1f ms, scoring + EER / minDCF %.1f ms" % (
        t_gpu / N_SUB * 1e3, t_prep / N_SUB * 1e3, t_val / N_SUB * 1e3), flush=True)
