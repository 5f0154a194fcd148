"""
GPU: the 1024-channel backbone family (reference tdnn16m4g / tdnn26m7g, sugar/models/dynamictdnn.py:942-948) on the HIP
path, against embeddings of the real reference (tests/golden/round2.npz, oracle/make_golden_round2.py).  These are the
"1024-channel TDNN convs" BASELINE.json's north_star names: first/last convs 1024 -> 1024 (K = 1024: the 256 x 256
ping-pong GEMM), catconv 3072 -> 1536, layer1 80 -> 1024 as the tap GEMM.  Res2 width 128 exceeds the fused chain
kernel's 64-channel tile, so those blocks run the Res2 steps as 7 generic launches (documented fallback).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sugar_amd
from sugar_amd import layout, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
WIDE = {"full": (3, [1024] * 4, [5] * 4, 1536), "sub": (2, [768, 1024, 512], [3, 5, 1], 1152)}
TOL = {"fp32": (0.9999, 1e-4), "bf16": (0.999, 3e-2)}          # cosine >=, relative L2 error <  (as test_gpu_parity)


@pytest.fixture(scope="module")
def wide_supernet():
    sup = sugar_amd.tdnn16m4g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(80, 192, 3, 1024, 5, 1536), seed=0))
    return sup


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "round2.npz"))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["full", "sub"])
def test_1024_channel_subnets_match_reference(name, precision, wide_supernet, gold):
    sub = wide_supernet.clone(WIDE[name]).set_precision(precision)
    wide_supernet.config = None
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()
    emb = model(synth.waveforms(4, 48000, seed=0).to(DEV)).double().cpu()
    want = torch.from_numpy(gold["wide/%s/4x3s" % name]).double()
    cos = F.cosine_similarity(emb, want, dim=1).min().item()
    rel = ((emb - want).norm(dim=1) / want.norm(dim=1)).max().item()
    print("wide", name, precision, "cosine min %.6f  rel L2 err max %.2e" % (cos, rel))
    assert cos >= TOL[precision][0] and rel < TOL[precision][1]


def test_1024_channel_batch_is_deterministic_and_batch_invariant(wide_supernet):
    """64 x 3 s through the 1024-channel supernet (R = 19 264 frames: full 256 x 256 rounds in every 1x1 conv)."""
    sub = wide_supernet.clone(WIDE["full"]).set_precision("bf16")
    wide_supernet.config = None
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()
    wav = synth.waveforms(4, 48000, seed=0)
    big = model(wav.repeat(16, 1).to(DEV))
    small = model(wav.to(DEV))
    assert torch.isfinite(big).all() and torch.equal(model(wav.repeat(16, 1).to(DEV)), big)
    assert torch.equal(big.view(16, 4, 192), small.unsqueeze(0).expand(16, 4, 192))


def test_1024_channel_persistent_gemm_is_bit_identical_to_one_workgroup_per_tile(wide_supernet):
    """176 x 3 s through the 1024-channel supernet: R = 52 976 frames = 828 tiles of 256 x 256 in the 1024 -> 1024 convs
    (three rounds + a split tail), so first_conv (with its duplicate store of the 8th Res2 split), last_conv and
    catconv all take the persistent GEMM; one workgroup per tile must give the same embeddings bit for bit."""
    from sugar_amd import _lib
    sub = wide_supernet.clone(WIDE["full"]).set_precision("bf16")
    wide_supernet.config = None
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()
    wav = synth.waveforms(4, 48000, seed=0).repeat(44, 1).to(DEV)
    persistent = model(wav)
    with _lib.tuned(gemm_no_persist=1):
        per_tile = model(wav)
    assert torch.isfinite(persistent).all() and torch.equal(persistent, per_tile)
    small = model(wav[:4])
    assert torch.equal(persistent.view(44, 4, 192), small.unsqueeze(0).expand(44, 4, 192))
