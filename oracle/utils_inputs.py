"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Seeded inputs shared by oracle/make_golden_utils.py and tests/test_utils.py."""
import numpy as np


def val_inputs():
    """(labels {0,1} [600], scores [600], a second score column)."""
    rs = np.random.RandomState(23)
    y = (rs.rand(600) < 0.4).astype(np.int64)
    s1 = rs.randn(600) * 0.25 + y * 0.6
    s2 = rs.randn(600) * 0.30 + y * 0.5
    return y, s1, s2


def wav_inputs():
    """name -> (audio, num_samples, noverlap)."""
    rs = np.random.RandomState(24)
    return {"1d": (rs.randn(1000).astype(np.float32), 300, None),
            "2d": (rs.randn(1, 777).astype(np.float32), 256, None),
            "exact": (rs.randn(512).astype(np.float32), 256, None)}
