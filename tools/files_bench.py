#!/usr/bin/env python3
"""
BASELINE config 4 FROM WAV FILES, the way a user of the reference runs it (sugar/vectors.py:126-153: file list ->
loadWAV -> forward -> score): N synthetic 16-bit PCM wav files of 2-20 s are written to a scratch directory, a trial
list file is written beside them, and

    trials = VerificationTrials(trials_lst, num_samples=0, mode_eval=True, num_eval=1, trials_dir=...)
    eer, dcf = veri_validate(loader(trials, batch_size=1, num_workers=J), model)

is timed end to end (file reads + decode + host packing + PCIe + fbank + TDNN + scoring + EER / minDCF) for J = 1 (the
serial walk) and J = 8 / 16 loader threads, next to the same list already resident in HBM.

    python tools/files_bench.py [n_files] [n_trials] [scratch_dir]
"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from scipy.io import wavfile  # noqa: E402

import sugar_amd  # noqa: E402
from sugar_amd import layout, synth  # noqa: E402
from sugar_amd.database import VerificationTrials  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4874
N_TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 37720
root = tempfile.mkdtemp(prefix="sugar_wavs_", dir=sys.argv[3] if len(sys.argv) > 3 else None)
DEV = "cuda:0"
try:
    rng = np.random.default_rng(7)
    lens = rng.integers(32000, 320001, size=N)
    t0 = time.perf_counter()
    for i, n in enumerate(lens):
        wavfile.write(os.path.join(root, "utt%05d.wav" % i), 16000, (rng.standard_normal(n) * 3000).astype(np.int16))
    nbytes = int(lens.sum()) * 2
    print("wrote %d wav files, %.2f GB of 16-bit PCM (%.0f s of audio) in %.1f s" % (N, nbytes / 1e9, lens.sum() / 16000,
                                                                                 time.perf_counter() - t0), flush=True)
    lab, a, b = synth.trial_pairs(N, N_TRIALS, seed=1)
    lst = os.path.join(root, "trials.txt")
    with open(lst, "w") as f:
        for i in range(N_TRIALS):
            f.write("%d utt%05d.wav utt%05d.wav\n" % (int(lab[i]), a[i], b[i]))
    trials = VerificationTrials(lst, num_samples=0, mode_eval=True, num_eval=1, trials_dir=root)

    class Loader:
        def __init__(self, dataset, batch_size, num_workers):
            self.dataset, self.batch_size, self.num_workers = dataset, batch_size, num_workers

    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    sub = sup.clone(layout.BASE).set_precision("bf16")
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()
    sugar_amd.veri_validate(Loader(trials, 1, 8), model, device=DEV)          # warm-up: page cache, weights, index memo
    t0 = time.perf_counter()
    for i in range(N):
        trials.utterances[i]
    print("serial walk alone (4 874 x loadWAV, page cache warm): %.3f s" % (time.perf_counter() - t0), flush=True)
    ref = None
    for jobs in (1, 4, 8, 12, 16, 4, 8, 12, 16):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eer, dcf, vec, scs = sugar_amd.veri_validate(Loader(trials, 1, jobs), model, device=DEV, ret_info=True)
        dt = time.perf_counter() - t0
        st = getattr(sugar_amd.extract_vectors, "last_hostfeed_stats", {})
        print("from files, %2d loader threads: %.3f s wall for %d utterances + %d trials (%.0f audio-s/s; producer: "
              "loading %.3f s, packing %.3f s, waiting for a slot %.3f s)  EER %.4f" % (
                  jobs, dt, N, N_TRIALS, lens.sum() / 16000 / dt, st.get("load_s", 0), st.get("pack_s", 0),
                  st.get("wait_slot_s", 0), eer[0]), flush=True)
        if ref is None:
            ref = (vec.matrix.clone(), scs.copy())
        else:
            print("   vectors / scores bit-equal to the first run: %s / %s" % (torch.equal(vec.matrix, ref[0]),
                                                                              bool((scs == ref[1]).all())), flush=True)
    packed = trials.utterances.stage(DEV)
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eer, dcf = sugar_amd.veri_validate(Loader(trials, 1, 0), model, device=DEV,
                                           vectors=sugar_amd.extract_vectors(model, packed))
        print("list resident in HBM: %.3f s wall" % (time.perf_counter() - t0), flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
