"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Fixture generator of round 2; runs ONLY in the build container, where
/root/reference exists (same import recipe as oracle/make_golden.py: thop / geatpy / torchaudio are absent from the
image and never called on this path, so they are registered as inert placeholders).

    PYTHONDONTWRITEBYTECODE=1 python3 -W ignore oracle/make_golden_round2.py

Writes tests/golden/round2.npz (outputs only; inputs regenerate from seeds):
  wide/*     the 1024-channel backbone family (reference tdnn16m4g, sugar/models/dynamictdnn.py:942-944): embeddings of
             the full supernet config and of one shrunk subnet for 4 x 3 s, from the reference's own clone + forward
  ill/*      Base and Largest under the ILL-CONDITIONED weight fill of sugar_amd/synth.py (saturated tanh, near one-hot
             softmax): the regime where the bf16 path is known to fall below the 0.999 cosine bar
  score/*    BASELINE config 4 at full size: the reference's calculate_score (sugar/scores.py:114-153) on 4 874 synthetic
             unit vectors and the 37 720-pair trial list
"""
import os
import sys
import types
from unittest.mock import MagicMock

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

sys.modules["thop"] = MagicMock()
_ea = types.ModuleType("geatpy")
_ea.Problem = object
sys.modules["geatpy"] = _ea
_ta = MagicMock()
sys.modules["torchaudio"] = _ta
sys.modules["torchaudio.transforms"] = _ta.transforms

import numpy as np  # noqa: E402
import torch  # noqa: E402

from sugar.models.dynamictdnn import tdnn8m2g, tdnn16m4g  # noqa: E402  (reference)
from sugar.scores import calculate_score  # noqa: E402  (reference)

from oracle import ref_cpu  # noqa: E402
from sugar_amd import layout, synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)

WIDE_FULL = (3, [1024] * 4, [5] * 4, 1536)
WIDE_SUB = (2, [768, 1024, 512], [3, 5, 1], 1152)
N_UTTS, N_TRIALS = 4874, 37720


def maxrel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    out = {}
    wav4 = synth.waveforms(4, 48000, seed=0)
    x301 = ref_cpu.logmel_fbank(wav4).squeeze(1)

    # ---- 1024-channel family ---------------------------------------------------------------------------------
    sup = tdnn16m4g(80, 192).eval()
    shapes = layout.supernet_shapes(80, 192, 3, 1024, 5, 1536)
    assert list(shapes.keys()) == list(sup.state_dict().keys())
    sd = synth.fill_state_dict(shapes, seed=0)
    sup.load_state_dict(sd, strict=True)
    for name, cfg in (("full", WIDE_FULL), ("sub", WIDE_SUB)):
        e = sup.clone(cfg).eval()(x301)
        o = ref_cpu.subnet_forward(ref_cpu.clone_state(sd, cfg, num_blocks=3, channels=1024), cfg[0], x301)
        assert maxrel(o, e) < 2e-5, (name, maxrel(o, e))
        out["wide/%s/4x3s" % name] = e.numpy()
        print("wide", name, "oracle vs reference", maxrel(o, e))

    # ---- ill-conditioned fill, Base and Largest --------------------------------------------------------------
    sup8 = tdnn8m2g(80, 192).eval()
    sd_ill = synth.fill_state_dict(layout.supernet_shapes(), seed=0, **synth.ILL_CONDITIONED)
    sup8.load_state_dict(sd_ill, strict=True)
    for name, cfg in (("base", layout.BASE), ("largest", layout.LARGEST)):
        e = sup8.clone(cfg).eval()(x301)
        o = ref_cpu.subnet_forward(ref_cpu.clone_state(sd_ill, cfg), cfg[0], x301)
        assert maxrel(o, e) < 5e-5, (name, maxrel(o, e))
        out["ill/%s/4x3s" % name] = e.numpy()
        print("ill", name, "oracle vs reference", maxrel(o, e))

    # ---- full-size cosine scoring (config 4) -----------------------------------------------------------------
    vec = torch.nn.functional.normalize(synth.normal("score4/vec", (N_UTTS, 192), seed=4), p=2, dim=1)
    lab, a, b = synth.trial_pairs(N_UTTS, N_TRIALS, seed=4)
    sc = np.array([calculate_score(vec[i:i + 1], vec[j:j + 1], "cosine") for i, j in zip(a.tolist(), b.tolist())])
    assert sc.dtype == np.float64 and sc.shape == (N_TRIALS,)
    mine = ref_cpu.score_trials(a, b, [vec[i:i + 1] for i in range(N_UTTS)])
    assert np.array_equal(sc, mine)
    out["score/full"] = sc
    gaps = np.diff(np.sort(sc))
    print("score/full: min gap %.2e, gaps below 1e-5: %d of %d" % (gaps.min(), int((gaps < 1e-5).sum()), N_TRIALS - 1))

    np.savez_compressed(os.path.join(GOLD, "round2.npz"), **out)
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(GOLD, "round2.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
