"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Fixture generator for the verification metrics; runs ONLY in the build
container, where /root/reference exists.

    PYTHONDONTWRITEBYTECODE=1 python3 -W ignore oracle/make_golden_metrics.py

Imports the reference's own sugar/metrics.py (calculate_eer :45-74, calculate_mindcf :180-218) with the inert thop
placeholder of make_golden.py, runs them on seeded synthetic trial scores and writes inputs-by-seed + outputs to
tests/golden/metrics.json.  `np.float` (removed from numpy >= 1.24, used at sugar/metrics.py:74) is aliased to the
builtin float for the duration of the run -- the reference file itself is not touched.
"""
import json
import os
import sys
from unittest.mock import MagicMock

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
sys.modules["thop"] = MagicMock()

import numpy as np  # noqa: E402

if not hasattr(np, "float"):
    np.float = float

from sugar.metrics import calculate_eer, calculate_mindcf  # noqa: E402  (reference)


from oracle.make_golden_metrics_inputs import trial_scores  # noqa: E402

CASES = [
    {"seed": 1, "n": 2000, "frac": 0.5, "sep": 2.0, "quant": None},
    {"seed": 2, "n": 37720, "frac": 0.5, "sep": 3.5, "quant": None},      # Vox1-O sized
    {"seed": 3, "n": 5000, "frac": 0.1, "sep": 1.0, "quant": None},
    {"seed": 4, "n": 3000, "frac": 0.3, "sep": 2.5, "quant": 8},          # heavy ties
    {"seed": 5, "n": 400, "frac": 0.5, "sep": 8.0, "quant": None},        # separable: EER 0
]


def main():
    out = []
    for c in CASES:
        y, s = trial_scores(c["seed"], c["n"], c["frac"], c["sep"], c["quant"])
        eer, thr = calculate_eer(y, s)
        rec = dict(c)
        rec["eer"] = float(eer)
        rec["eer_thresh"] = float(thr)
        for p in (0.01, 0.05):
            d, t = calculate_mindcf(y, s, p_target=p)
            rec["mindcf_%g" % p] = float(d)
            rec["mindcf_thresh_%g" % p] = float(t)
        out.append(rec)
        print(rec)
    with open(os.path.join(ROOT, "tests", "golden", "metrics.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
