"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Fixture generator for calculate_score(mode="norm2") -- the reference's default mode
-- and for list-of-modes scoring; runs ONLY in the build container, where /root/reference exists.

    PYTHONDONTWRITEBYTECODE=1 python3 -W ignore oracle/make_golden_norm2.py

Runs the reference's own calculate_score / score_trials (sugar/scores.py:114-216) on seeded vectors, asserts the oracle
restatement (oracle/ref_cpu.py::norm2_score) reproduces them, and writes the outputs to tests/golden/norm2.npz.
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

from sugar.scores import calculate_score, score_trials  # noqa: E402  (reference)

from oracle import ref_cpu  # noqa: E402
from oracle.make_golden_norm2_inputs import inputs  # noqa: E402


def main():
    vectors, lab, a, b = inputs()
    table = {"utt%03d" % i: v for i, v in enumerate(vectors)}
    trials = [{"label": int(lab[i]), "enroll": "utt%03d" % a[i], "test": "utt%03d" % b[i]} for i in range(len(lab))]
    out = {}
    norm2 = np.array([calculate_score(table[t["enroll"]], table[t["test"]], mode="norm2") for t in trials])
    assert np.allclose(norm2, ref_cpu.score_trials(a, b, vectors, mode="norm2"), rtol=0, atol=1e-6)
    out["norm2"] = norm2
    both, labels, pairs = score_trials(trials, table, mode=["cosine", "norm2"], n_jobs=1)
    assert both.shape == (len(trials), 2) and np.array_equal(both[:, 1], norm2)
    assert np.allclose(both, ref_cpu.score_trials(a, b, vectors, mode=["cosine", "norm2"]), rtol=0, atol=1e-6)
    out["cosine_norm2"] = both
    out["labels"] = labels
    default = calculate_score(vectors[1], vectors[2])                 # the reference's default mode IS norm2
    assert default == calculate_score(vectors[1], vectors[2], mode="norm2")
    out["default_1_2"] = np.array(default)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "norm2.npz"), **out)
    print({k: v.shape for k, v in out.items()}, norm2[:4])


if __name__ == "__main__":
    main()
