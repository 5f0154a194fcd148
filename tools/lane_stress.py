#!/usr/bin/env python3
"""Determinism of extract_vectors under stream-level overlap: the same packed list embedded N times on LANES concurrent
HIP streams must give bit-identical vectors every time (one stream is the reference).  This is the test that exposed
wrong packed-FP32 results in a wave whose CU also ran MFMA waves of another kernel; the library
is built without packed-FP32 instructions since then."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sugar_amd  # noqa: E402
import sugar_amd.vectors as V  # noqa: E402
from sugar_amd import layout, synth  # noqa: E402
from sugar_amd.vectors import PackedUtterances, extract_vectors  # noqa: E402

dev = torch.device("cuda", 0)
sup = sugar_amd.tdnn8m2g()
sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
n_utts = 48
lens = synth.ragged_lengths(n_utts, lo=24000, hi=40000, seed=13).tolist()
wavs = synth.ragged_waveforms(lens, seed=13)


class Utts:
    def __len__(self):
        return n_utts

    def __getitem__(self, i):
        return {"file": "utt%04d.wav" % i, "input": wavs[i].unsqueeze(0)}


packed = PackedUtterances(Utts(), dev, batch_size=4)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lanes = int(os.environ.get("LANES", "4"))
total_bad = 0
for cfg in ((3, [176, 384, 128, 176], [3, 5, 1, 3], 536), layout.BASE, (2, [128, 128, 128], [1, 1, 1], 384)):
    sub = sup.clone(cfg).set_precision("bf16")
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).eval()
    V.N_LANES = 1
    ref = extract_vectors(model, packed).matrix.clone()
    V.N_LANES = lanes
    bad = 0
    for it in range(N):
        bad += int(not torch.equal(extract_vectors(model, packed).matrix, ref))
    total_bad += bad
    print("config %s: %d of %d runs on %d lanes differ from the one-stream result" % (cfg[1], bad, N, lanes), flush=True)
sys.exit(1 if total_bad else 0)
