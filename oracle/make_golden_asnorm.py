"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Fixture generator for the score-normalisation back-end; runs ONLY in the build
container, where /root/reference exists.

    PYTHONDONTWRITEBYTECODE=1 python3 -W ignore oracle/make_golden_asnorm.py

Runs the reference's own score_cohorts / asnorm / snorm (sugar/scores.py:77-111, 269-400) on seeded inputs, asserts the
oracle restatement (oracle/ref_cpu.py) reproduces them, and writes the outputs to tests/golden/asnorm.npz.
Same inert placeholders for thop / geatpy / torchaudio as make_golden.py.
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
import pandas as pd  # noqa: E402

from sugar.scores import asnorm, score_cohorts, snorm  # noqa: E402  (reference)

from oracle import ref_cpu  # noqa: E402
from oracle.make_golden_asnorm_inputs import NTOPS, inputs  # noqa: E402


def main():
    vectors, cohorts, scores, enr, tst = inputs()
    out = {}
    cs = score_cohorts(cohorts, vectors, device="cpu")
    ocs = ref_cpu.score_cohorts(cohorts, vectors)
    keys = list(vectors.keys())
    out["cohort_scores"] = np.stack([cs[k] for k in keys])
    assert np.array_equal(out["cohort_scores"], np.stack([ocs[k] for k in keys]))
    df = pd.DataFrame({"score": scores, "enroll": enr, "test": tst})
    for nt in NTOPS:
        ref = np.asarray(asnorm(df, cs, nTop=nt))
        assert np.allclose(ref, ref_cpu.asnorm(scores, enr, tst, ocs, nt), rtol=1e-6, atol=1e-6), nt
        out["asnorm_%d" % nt] = ref
    ref = np.asarray(snorm(df, cs))
    assert np.allclose(ref, ref_cpu.snorm(scores, enr, tst, ocs), rtol=1e-6, atol=1e-6)
    out["snorm"] = ref
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "asnorm.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
