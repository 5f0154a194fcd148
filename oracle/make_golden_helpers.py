"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Fixture generator for the small helpers on sugar's scoring / vector call surface
(round 3 widening): ztnorm, score_batch, normalize_scores, save_sctr / load_sctr (sugar/scores.py:27-74, 219-266,
402-420) and length_normalize / vector_transform (sugar/vectors.py:25-60, 157-178).  Runs ONLY in the build container.

    PYTHONDONTWRITEBYTECODE=1 python3 -W ignore oracle/make_golden_helpers.py

Inputs are the asnorm fixture's (oracle/make_golden_asnorm_inputs.py); outputs of the reference's own functions go to
tests/golden/helpers.npz.  Same inert placeholders for thop / geatpy / torchaudio as make_golden.py.
"""
import os
import sys
import tempfile
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
import pandas as pd  # noqa: E402
import torch  # noqa: E402

from sugar.scores import load_sctr, normalize_scores, save_sctr, score_batch, score_cohorts, ztnorm  # noqa: E402
from sugar.vectors import length_normalize, vector_transform  # noqa: E402

from oracle.make_golden_asnorm_inputs import inputs  # noqa: E402


def main():
    vectors, cohorts, scores, enr, tst = inputs()
    out = {}
    cs = score_cohorts(cohorts, vectors, device="cpu")
    out["cohort_scores"] = np.stack([cs[k] for k in vectors])       # the ztnorm inputs, as the reference computed them here
    df = pd.DataFrame({"score": scores, "enroll": enr, "test": tst})
    for nt in (None, 300, -300):
        out["ztnorm_%s" % nt] = np.asarray(ztnorm(df, cs, nTop=nt), dtype=np.float64)
    out["score_batch"] = score_batch(vectors["utt002.wav"], cohorts[:50])            # [n_seg, 50]
    out["normalize_scores"] = normalize_scores(scores, -1.0, 2.0)
    v = vectors["utt005.wav"]
    for su in (False, True):
        n, avg, std = length_normalize(v.clone(), scaleup=su)
        out["lennorm_%d" % su] = n.numpy()
        out["lennorm_%d_stats" % su] = np.array([float(avg), float(std)])
    n, avg, std = length_normalize(v.numpy().copy(), scaleup=True)
    out["lennorm_np"] = n
    out["lennorm_np_stats"] = np.array([float(avg), float(std)])
    vt = vector_transform({k: vectors[k] for k in list(vectors)[:3]}, transform=lambda x: x * 2.0)
    out["vector_transform"] = np.concatenate([vt[k].numpy() for k in vt])
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scores.txt")
        pairs = [[e, t] for e, t in zip(enr, tst)]
        save_sctr(scores[:20], pairs[:20], path, root_dir="utt", scale=True)
        out["sctr_text"] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        sc, tr = load_sctr(path)
        out["sctr_scores"] = np.array(sc)
        assert tr[0] == [pairs[0][0].replace("utt", ""), pairs[0][1].replace("utt", "")]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "helpers.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
