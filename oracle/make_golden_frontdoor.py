"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Fixture generator for the data front door; runs ONLY in the build container.

    PYTHONDONTWRITEBYTECODE=1 python3 -W ignore oracle/make_golden_frontdoor.py

Runs the reference's own loadWAV (sugar/waveforms.py:47-115), Utterance.collate_fn and VerificationTrials
(sugar/database.py:105-158, 289-364) on seeded wav files / a trial list written to a temporary directory and stores
what they return in tests/golden/frontdoor.json (paths relative to that directory; arrays as shape + SHA-256).
Same inert placeholders for thop / geatpy / torchaudio as make_golden.py.
"""
import hashlib
import json
import os
import random
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

from sugar.database import Utterance, VerificationTrials, txt2list  # noqa: E402  (reference)
from sugar.waveforms import loadWAV  # noqa: E402  (reference)

from oracle.make_golden_frontdoor_inputs import LOAD_CASES, WAVS, write_files  # noqa: E402


class _Digests(dict):
    """Outputs are compared exactly, so the fixture keeps shape + SHA-256 of the float32 bytes (the arrays
    themselves would be 6 MB)."""

    def __setitem__(self, k, v):
        v = np.ascontiguousarray(v, dtype=np.float32)
        dict.__setitem__(self, k, [list(v.shape), hashlib.sha256(v.tobytes()).hexdigest()])


def main():
    out, meta = _Digests(), {}
    with tempfile.TemporaryDirectory() as root:
        write_files(root)
        for ci, (ns, ev, ne, mn, seed) in enumerate(LOAD_CASES):
            for name in WAVS:
                random.seed(seed)
                out["load/%d/%s" % (ci, name)] = loadWAV(os.path.join(root, name), ns, ev, ne, 16000, mn)
        random.seed(5)
        tr = VerificationTrials(os.path.join(root, "trials.txt"), num_samples=0, mode_eval=True, num_eval=1, trials_dir=root)
        rel = lambda p: os.path.relpath(p, root)
        meta["labels"] = [int(x) for x in tr.labels]
        meta["enrolls"] = [rel(p) for p in tr.enrolls]
        meta["tests"] = [rel(p) for p in tr.tests]
        meta["files"] = [rel(p) for p in tr.files]
        meta["item2"] = {k: (rel(v) if isinstance(v, str) else int(v)) for k, v in tr[2].items()}
        it = tr.utterances[0]
        meta["utt0_file"] = rel(it["file"])
        out["utt0_input"] = it["input"].numpy()
        ds = Utterance(list(WAVS), num_samples=16000, mode_eval=True, num_eval=3, root_dir=root)
        batch = ds.collate_fn([ds[0], ds[2]])
        meta["collate_files"] = [rel(p) for p in batch["file"]]
        meta["collate_num"] = [int(n) for n in batch["num"]]
        out["collate_input"] = batch["input"].numpy()
        meta["txt2list"] = [rel(p) for p in txt2list(os.path.join(root, "utts.txt"), root)]
    meta["arrays"] = dict(out)
    with open(os.path.join(ROOT, "tests", "golden", "frontdoor.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(meta)


if __name__ == "__main__":
    main()
