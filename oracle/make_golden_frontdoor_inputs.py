"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Seeded wav files / trial lists shared by oracle/make_golden_frontdoor.py and
tests/test_frontdoor.py (written to a temporary directory on both sides, never stored)."""
import os

import numpy as np
from scipy.io import wavfile

WAVS = {"spk1/a.wav": 20000, "spk1/b.wav": 48000, "spk2/c.wav": 130001, "spk2/d.wav": 47999}
# (num_samples, mode_eval, num_eval, min_samples, random seed)
LOAD_CASES = [(48000, False, 6, 48000, 7), (48000, True, 6, 48000, 0), (0, True, 1, 48000, 0), (32000, True, 3, None, 0),
              (16000, False, 2, 48000, 123), (64000, True, 2, 48000, 0)]
TRIALS = ["1 spk1/a.wav spk1/b.wav", "0 spk1/a.wav spk2/c.wav", "spk2/d.wav spk1/b.wav", "0 spk2/c.wav spk1/b.wav",
          "1 spk2/c.wav spk2/d.wav"]


def write_files(root):
    for i, (name, n) in enumerate(WAVS.items()):
        os.makedirs(os.path.dirname(os.path.join(root, name)), exist_ok=True)
        rs = np.random.RandomState(40 + i)
        wavfile.write(os.path.join(root, name), 16000, (rs.randn(n) * 3000).astype(np.int16))
    with open(os.path.join(root, "trials.txt"), "w") as f:
        f.write("\n".join(TRIALS) + "\n")
    with open(os.path.join(root, "utts.txt"), "w") as f:
        f.write("spk1 spk1/a.wav\nspk2/c.wav\n")
