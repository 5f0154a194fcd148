"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Seeded synthetic trial lists shared by the metrics fixture generator
(oracle/make_golden_metrics.py) and tests/test_metrics.py; inputs are regenerated from seeds, never stored."""
import numpy as np


def trial_scores(seed, n, p_target_frac, sep, quant=None):
    """labels ~ Bernoulli(p_target_frac), scores = N(sep * label, 1), optionally quantised to 1/quant (ties)."""
    rs = np.random.RandomState(seed)
    y = (rs.rand(n) < p_target_frac).astype(np.int64)
    s = rs.randn(n) + sep * y
    if quant:
        s = np.round(s * quant) / quant
    return y, s
