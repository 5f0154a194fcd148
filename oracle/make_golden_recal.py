"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Fixture generator for BN recalibration; runs ONLY in the build container,
where /root/reference exists.

    PYTHONDONTWRITEBYTECODE=1 python3 -W ignore oracle/make_golden_recal.py

Runs the real reference subnet (sugar.models.dynamictdnn, cloned with the reference's own `clone`) the way
`batch_forward` does (sugar/models/__init__.py:230-238: `model.train()`, forward over the batches, no grad), then
`eval()` and one more forward.  Asserts that oracle/ref_cpu.py (train_momentum=0.1) reproduces the running
statistics of every BatchNorm and the outputs, and writes them to tests/golden/recal.npz (data only: waveforms
regenerate from seeds; the front-end is the oracle's -- PARITY UNPINNED for that stage, see ref_cpu.py).
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
import torch  # noqa: E402

from sugar.models.dynamictdnn import tdnn8m2g  # noqa: E402  (reference)

from oracle import ref_cpu  # noqa: E402
from sugar_amd import layout, synth  # noqa: E402

torch.set_grad_enabled(False)

from oracle.make_golden_recal_inputs import CASES, EVAL_SEED, batches  # noqa: E402


def main():
    sup = tdnn8m2g(80, 192).eval()
    sup_sd = synth.fill_state_dict(layout.supernet_shapes(), seed=0)
    sup.load_state_dict(sup_sd, strict=True)
    out = {}
    for name, (cfg, B, L, nb) in CASES.items():
        sub = sup.clone(cfg)
        osd = ref_cpu.clone_state(sup_sd, cfg)
        sub.train()                                                   # batch_forward: model.train()
        for bi, wav in enumerate(batches(name, nb, B, L)):
            feats = ref_cpu.logmel_fbank(wav).squeeze(1)
            e_ref = sub(feats)
            e_or = ref_cpu.subnet_forward(osd, cfg[0], feats, train_momentum=0.1)
            assert torch.allclose(e_ref, e_or, rtol=1e-4, atol=1e-5), (name, bi, (e_ref - e_or).abs().max())
            out["%s/train_emb%d" % (name, bi)] = e_ref.numpy()
        sub.eval()
        ssd = sub.state_dict()
        for k, v in ssd.items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                assert torch.allclose(v, osd[k], rtol=1e-5, atol=1e-6), (name, k, (v - osd[k]).abs().max())
                out["%s/%s" % (name, k)] = v.numpy()
        wav = synth.waveforms(3, L, seed=EVAL_SEED)
        feats = ref_cpu.logmel_fbank(wav).squeeze(1)
        e_ref = sub(feats)
        e_or = ref_cpu.subnet_forward(osd, cfg[0], feats)
        assert torch.allclose(e_ref, e_or, rtol=1e-4, atol=1e-5), (name, "eval")
        out["%s/eval_emb" % name] = e_ref.numpy()
        print(name, "ok:", sum(1 for k in out if k.startswith(name + "/")), "arrays")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "recal.npz"), **out)


if __name__ == "__main__":
    main()
