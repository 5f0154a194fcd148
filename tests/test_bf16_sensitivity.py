"""
GPU: where the bf16 path crosses BASELINE's 0.999 cosine bar.

Round 1 met the bar on a weight fill that had been re-conditioned (DESIGN.md "bf16 sensitivity"): with the first fill
(conv gain sqrt(2), pooling.linear2 x 4) tanh saturates and the softmax over time is nearly one-hot, and the pooled
statistics become a discontinuous function of the activations.  That fill is kept here as a fixture
(synth.ILL_CONDITIONED; the REAL reference's embeddings for it are in tests/golden/round2.npz), and the sweep walks the
geometric path from the shipped fill (t = 0) to it (t = 1) for Base and Largest:
  * the exact-f32 path must stay >= 0.9999 EVERYWHERE (it is the parity path; conditioning must not matter to it);
  * the bf16 path must be >= 0.999 at t = 0 and where it stands along the path is RECORDED, not hidden: the table goes
    to gpurun_out/bf16_sensitivity.log (copied to profiles/r02_bf16_sensitivity.log by the builder).  Measured in round
    2: the cosine bar holds all the way (Largest 0.999051 at t = 1, Base 0.999983) but the relative L2 error of an
    embedding grows from 2e-3 to 4.4e-2 on Largest and crosses test_gpu_parity's 3e-2 bound between t = 0.875 and 1:
    at t = 1 the bf16 path is only just inside the cosine bar.  The test asserts the bar at t = 0 and a floor of 0.99
    at t = 1 (so that a regression of the ill-conditioned regime is seen) -- not more than the arithmetic supports.
The oracle (CPU restatement, pinned to the reference at t = 0 and t = 1 by the goldens) provides the in-between points.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sugar_amd
from oracle import ref_cpu
from sugar_amd import layout, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {"base": layout.BASE, "largest": layout.LARGEST}


def _embed(sd, cfg, precision, wav):
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(sd)
    sub = sup.clone(cfg).set_precision(precision)
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()
    emb = model(wav.to(DEV)).double().cpu()
    sub.release()
    return emb


def _cmp(emb, want):
    want = torch.as_tensor(want).double()
    return (F.cosine_similarity(emb, want, dim=1).min().item(), ((emb - want).norm(dim=1) / want.norm(dim=1)).max().item())


def test_ill_conditioned_fixture_f32_exact_bf16_recorded():
    gold = np.load(os.path.join(ROOT, "tests", "golden", "round2.npz"))
    sd = synth.fill_state_dict(layout.supernet_shapes(), seed=0, **synth.ILL_CONDITIONED)
    wav = synth.waveforms(4, 48000, seed=0)
    for name, cfg in CONFIGS.items():
        c32, r32 = _cmp(_embed(sd, cfg, "fp32", wav), gold["ill/%s/4x3s" % name])
        c16, r16 = _cmp(_embed(sd, cfg, "bf16", wav), gold["ill/%s/4x3s" % name])
        print("ill-conditioned %s: f32 cosine %.6f rel %.1e | bf16 cosine %.6f rel %.1e" % (name, c32, r32, c16, r16))
        assert c32 >= 0.9999 and r32 < 1e-3          # activations x130: the f32 path's own rounding is still ~1e-5
        assert c16 >= 0.99                           # measured 0.999983 / 0.999051: marginal on Largest (round 1: 0.9963)


def test_gain_sweep_records_where_bf16_crosses_the_bar():
    wav = synth.waveforms(4, 48000, seed=0)
    ts = [0.0, 0.25, 0.5, 0.625, 0.75, 0.875, 1.0]
    lines = ["# bf16 / f32 path vs the CPU oracle on 4 x 3 s, weight fill on the path shipped (t=0) -> ill-conditioned (t=1)",
             "# t  conv_gain pool1_gain pool2_gain | config  f32_cos_min f32_relL2 | bf16_cos_min bf16_relL2"]
    cross = {}
    for name, cfg in CONFIGS.items():
        prev_ok = True
        for t in ts:
            g = synth.conditioning(t)
            sd = synth.fill_state_dict(layout.supernet_shapes(), seed=0, **g)
            want = ref_cpu.embed(ref_cpu.clone_state(sd, cfg), cfg[0], wav)
            c32, r32 = _cmp(_embed(sd, cfg, "fp32", wav), want)
            c16, r16 = _cmp(_embed(sd, cfg, "bf16", wav), want)
            lines.append("%.3f %.4f %.4f %.4f | %-7s %.6f %.2e | %.6f %.2e" % (
                t, g["conv_gain"], g["pool1_gain"], g["pool2_gain"], name, c32, r32, c16, r16))
            assert c32 >= 0.9999, (name, t, c32)
            if t == 0.0:
                assert c16 >= 0.999, (name, c16)
            if prev_ok and c16 < 0.999:
                cross[name] = t
                prev_ok = False
        lines.append("# %s: bf16 cosine first below 0.999 at t = %s" % (name, cross.get(name, "never (<= 1.0)")))
    text = "\n".join(lines)
    print(text)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bf16_sensitivity.log"), "w") as f:
        f.write(text + "\n")
