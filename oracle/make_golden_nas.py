"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Fixture generator for the NAS tooling of SURVEY 8f rank 4 (ArchCoder, config2d,
AccuracyPredictor.predict, MinimizeError.encode: sugar/models/dynamictdnn.py:951-1172, 1505-1529); runs ONLY in the
build container, where /root/reference exists (same inert placeholders for thop / geatpy / torchaudio as
oracle/make_golden.py).

    PYTHONDONTWRITEBYTECODE=1 python3 -W ignore oracle/make_golden_nas.py

Writes tests/golden/nas.npz: what the reference's own classes return on seeded inputs -- one-hot / binary / continuous
encodings of 32 sampled coarse-space subnets, the predictions of a seeded AccuracyPredictor (its parameters are stored
as data: they come from torch's initialiser, not from a reference file) and MinimizeError.encode on random genomes.
geatpy is absent, so `ea_solve` itself cannot be run: the search loop of sugar_amd/nas.py is PARITY UNPINNED and is
tested against exhaustive search instead.  ArchCoder.decode of the reference uses `np.int` and raises on numpy >= 1.24
(recorded in `decode_raises`); decode is pinned by round trips.
"""
import os
import random
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

from sugar.models.dynamictdnn import AccuracyPredictor, ArchCoder, MinimizeError, config2d  # noqa: E402  (reference)

CSPACE = [[2, 3, 4]] + [[128, 176, 256, 384, 512]] * 5 + [[1, 3, 5]] * 5 + [[384, 536, 768, 1152, 1536]]
DIMS = ["depth"] + ["width%d" % i for i in range(5)] + ["kernel%d" % i for i in range(5)] + ["cat"]


def sample_configs(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        d = rng.choice(CSPACE[0])
        out.append([d] + [rng.choice(CSPACE[1]) for _ in range(d + 1)] + [rng.choice(CSPACE[6]) for _ in range(d + 1)]
                   + [rng.choice(CSPACE[11])])
    return out


def main():
    out = {}
    dconfig = {d: CSPACE[i] for i, d in enumerate(DIMS)}
    configs = sample_configs(32, seed=5)
    out["configs"] = np.array([c + [-1] * (12 - len(c)) for c in configs])
    for mode in (None, "b", "c"):
        codec = ArchCoder(len(CSPACE), dconfig, config2d, mode=mode)
        enc = np.stack([codec.encode(c) for c in configs])
        out["encode/%s" % (mode or "onehot")] = enc
        out["nfeat/%s" % (mode or "onehot")] = np.array(codec.nfeat)
    try:
        ArchCoder(len(CSPACE), dconfig, config2d).decode(out["encode/onehot"][0], 1)
        out["decode_raises"] = np.array(0)
    except AttributeError:
        out["decode_raises"] = np.array(1)
    d0 = config2d(configs[0])
    out["config2d_keys"] = np.array(list(d0.keys()))
    out["config2d_vals"] = np.array([-1 if v is None else v for v in d0.values()])

    # seeded predictor: parameters as data, predictions of the reference's own predict()
    torch.manual_seed(0)
    nfeat = int(out["nfeat/onehot"])
    pred = AccuracyPredictor(nfeat, device="cpu")
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        bn = pred.model[0]
        bn.running_mean.copy_(torch.rand(nfeat, generator=g) * 0.5)
        bn.running_var.copy_(torch.rand(nfeat, generator=g) * 0.3 + 0.05)
        bn.weight.copy_(torch.rand(nfeat, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(nfeat, generator=g) * 0.1)
        pred.ynorm.running_mean.fill_(3.1)
        pred.ynorm.running_var.fill_(0.37)
        pred.ynorm.weight.fill_(1.2)
        pred.ynorm.bias.fill_(-0.05)
    for k, v in pred.model.state_dict().items():
        out["pred/model/" + k] = v.numpy()
    for k, v in pred.ynorm.state_dict().items():
        out["pred/ynorm/" + k] = v.numpy()
    out["pred/y"] = pred.predict(out["encode/onehot"])
    out["pred/y_single"] = np.array(pred.predict(out["encode/onehot"][:1]))

    # MinimizeError.encode (genome -> flat config list), both spaces; __init__ needs geatpy, encode does not
    rng = np.random.default_rng(3)
    for mode, ub in (("Coarse", [4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 4]), ("Fine", [4, 48, 48, 48, 48, 48, 3, 3, 3, 3, 3, 144])):
        lb = [2, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 0]
        X = np.stack([rng.integers(lb[i], ub[i] + 1, 16) for i in range(12)], 1)
        me = types.SimpleNamespace(space_mode=mode)
        enc = MinimizeError.encode(me, X.copy())
        out["genome/%s" % mode] = X
        out["genome_encoded/%s" % mode] = np.array([e + [-1] * (12 - len(e)) for e in enc])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "nas.npz"), **out)
    print("wrote tests/golden/nas.npz;", {k: getattr(v, "shape", None) for k, v in out.items() if not k.startswith("pred/model")})


if __name__ == "__main__":
    main()
