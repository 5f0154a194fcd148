"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Fixture generator for host utilities next to the path (round 3 widening):
calculate_val (sugar/metrics.py:221-281), wav_segment (sugar/waveforms.py:118-146), txt2dict (sugar/database.py:39-69),
bn_state_dict / load_bn_state_dict / keep_params (sugar/utils/utility.py:9-42, 67-75).  Runs ONLY in the build container.

    PYTHONDONTWRITEBYTECODE=1 python3 -W ignore oracle/make_golden_utils.py

Outputs of the reference's own functions on seeded inputs -> tests/golden/utils.npz.  Same inert placeholders for thop /
geatpy / torchaudio as make_golden.py.
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
import torch  # noqa: E402

from sugar.database import txt2dict  # noqa: E402
from sugar.metrics import calculate_val  # noqa: E402
from sugar.models.dynamictdnn import tdnn8m2g  # noqa: E402
from sugar.utils.utility import bn_state_dict, keep_params, load_bn_state_dict  # noqa: E402
from sugar.waveforms import wav_segment  # noqa: E402

from oracle.utils_inputs import val_inputs, wav_inputs  # noqa: E402
from sugar_amd import layout, synth  # noqa: E402


def main():
    out = {}
    y, s1, s2 = val_inputs()
    for far in (1e-1, 1e-2):
        out["val_%g" % far] = np.array(calculate_val(y, s1, far=far), dtype=np.float64)
        out["val2d_%g" % far] = np.array(calculate_val(y, np.stack([s1, s2], axis=1), far=far), dtype=np.float64)
    for name, (audio, n, nov) in wav_inputs().items():
        out["wavseg_" + name] = wav_segment(audio, n, nov)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "utts.txt")
        with open(p, "w") as f:
            f.write("id01 a/1.wav\nid02 b/1.wav\nid01 a/2.wav\n")
        got = txt2dict(p, "/data")
        assert got == {"id01": ["/data/a/1.wav", "/data/a/2.wav"], "id02": ["/data/b/1.wav"]}, got
        # keep_params: what stays in a *.torchparams file
        ck = os.path.join(d, "latest.pth.tar")
        torch.save({"epoch": 3, "train_step": 7, "arch": "x", "state_dict": {"w": torch.ones(2)}, "optimizer": {"o": 1},
                    "scheduler": {"s": 2}, "lr_step": "epoch"}, ck)
        keep_params(ck, os.path.join(d, "p.torchparams"))
        kept = torch.load(os.path.join(d, "p.torchparams"), "cpu", weights_only=True)
        assert sorted(kept) == ["arch", "epoch", "state_dict", "train_step"], sorted(kept)
    # bn_state_dict / load_bn_state_dict round trip on a cloned reference subnet: the key list is the fixture
    sup = tdnn8m2g(80, 192).eval()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    sub = sup.clone(layout.SMALLEST).eval()
    bn = bn_state_dict(sub)
    out["bn_keys"] = np.array("\n".join(bn.keys()))
    out["bn_sum"] = np.array(sum(float(v.double().sum()) for v in bn.values()))
    sub2 = sup.clone(layout.SMALLEST).eval()
    for v in bn_state_dict(sub2).values():
        v.zero_()
    load_bn_state_dict(sub2, bn)
    assert all(torch.equal(a, b) for a, b in zip(bn_state_dict(sub2).values(), bn.values()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "utils.npz"), **out)
    print({k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
