"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Fixture generator for LONG utterances; runs ONLY in the build container, where
/root/reference exists (same placeholder arrangement as oracle/make_golden.py, which see).

    PYTHONDONTWRITEBYTECODE=1 python3 -W ignore oracle/make_golden_long.py

Utterances of 6.9 / 11 / 21 / 1 s (687, 1101, 2126 and 101 frames) are run ONE AT A TIME and unpadded through the
reference's own cloned Base and Largest subnets (its whole-utterance evaluation mode, sugar/waveforms.py:105-113
with num_samples=0) -- the lengths at which this repository cuts an utterance into time segments (Res2 chain > 384
frames, instance norm > 320, attentive pooling > 1024).  The oracle is asserted against the reference on the same
features, then the reference's embeddings are written to tests/golden/long.npz (outputs only; the inputs regenerate
from seeds).  Features come from oracle.ref_cpu.logmel_fbank (front-end PARITY UNPINNED, see ref_cpu.py).
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

from sugar.models.dynamictdnn import tdnn8m2g  # noqa: E402  (reference)

from oracle import ref_cpu  # noqa: E402
from sugar_amd import layout, synth  # noqa: E402

LONG_LENGTHS = [109840, 176000, 340001, 16000]
LONG_SEED = 21


def main():
    torch.set_grad_enabled(False)
    sup = tdnn8m2g(80, 192).eval()
    sup_sd = synth.fill_state_dict(layout.supernet_shapes(), seed=0)
    sup.load_state_dict(sup_sd, strict=True)
    wavs = synth.ragged_waveforms(LONG_LENGTHS, seed=LONG_SEED)
    out = {"lengths": np.array(LONG_LENGTHS), "seed": np.array(LONG_SEED)}
    for name, cfg in (("base", layout.BASE), ("largest", layout.LARGEST)):
        sub = sup.clone(cfg).eval()
        osd = ref_cpu.clone_state(sup_sd, cfg)
        embs = []
        for wv in wavs:
            xx = ref_cpu.logmel_fbank(wv.unsqueeze(0)).squeeze(1)
            embs.append(sub(xx))
        e = torch.cat(embs)
        o = ref_cpu.embed_ragged(osd, cfg[0], wavs)
        rel = float((o - e).abs().max() / e.abs().max())
        assert rel < 2e-5, (name, rel)
        print(name, "oracle vs reference max rel %.2e" % rel)
        out["embed/%s/long" % name] = e.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "long.npz"), **out)
    print("wrote tests/golden/long.npz")


if __name__ == "__main__":
    main()
