"""
GPU: the 1024-channel backbone family (reference tdnn16m4g / tdnn26m7g, sugar/models/dynamictdnn.py:942-948) on the HIP
path, against embeddings of the real reference (tests/golden/round2.npz, oracle/make_golden_round2.py).  These are the
"1024-channel TDNN convs" BASELINE.json's north_star names: first/last convs 1024 -> 1024 (K = 1024: the 256 x 256
ping-pong GEMM), catconv 3072 -> 1536, layer1 80 -> 1024 as the tap GEMM.  Res2 groups of 96 / 128 channels run in
res2_wide_kernel (the chain in one kernel, weights staged in 32- or 64-channel sub-passes); other widths above 64 fall
back to 7 generic launches per block.
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
    for sched in (1, 2):                             # (K = 1024: 16 k-tiles; with its duplicate store the prefetch wait is uncounted)
        with _lib.tuned(gemm_no_quad=sched):
            assert torch.equal(persistent, model(wav)), sched
    small = model(wav[:4])
    assert torch.equal(persistent.view(44, 4, 192), small.unsqueeze(0).expand(44, 4, 192))


def _row_off(frames):
    off = [0]
    for t in frames:
        off.append(off[-1] + t)
    return off


# 128-channel Res2 groups with 5 taps (32-channel weight sub-passes), with 3 taps (64-channel sub-passes), 96-channel groups
R2_WIDE_CASES = {"w128k5": (2, [1024, 1024, 1024], [5, 5, 5], 1536), "w128k3": (2, [1024, 1024, 1024], [5, 3, 3], 1536),
                 "w96": (2, [512, 768, 768], [3, 3, 5], 1536), "w128k1+w64": (2, [768, 1024, 512], [3, 1, 3], 1152)}


@pytest.mark.parametrize("name", sorted(R2_WIDE_CASES))
def test_wide_res2_chain_matches_per_step_path_bf16(name, wide_supernet):
    """The one-kernel Res2 chain for 65..128-channel groups (res2_wide_kernel) against the 7-launch generic path, same
    bf16 inputs; ragged utterances including ones cut into time segments.  As for the narrow kernel the two differ by
    one bf16 rounding per step (the fused chain adds x_j + sp_{j-1} before rounding)."""
    from sugar_amd import _lib
    cfg = R2_WIDE_CASES[name]
    sub = wide_supernet.clone(cfg).set_precision("bf16")
    wide_supernet.config = None
    frames = [301, 63, 17, 130, 700, 385, 16]
    R = sum(frames)
    feats = synth.normal("res2wide/feats", (R, 80), seed=9).to(torch.bfloat16).to(DEV)
    row_off = torch.tensor(_row_off(frames), dtype=torch.int32, device=DEV)
    depth = cfg[0]
    outs = {}
    for mode in ("fused", "generic"):
        with _lib.tuned(generic_res2=int(mode == "generic")):
            outs[mode] = [sub.tap(feats, row_off, len(frames), R, max(frames), 2 + 4 * i)[0].clone() for i in range(depth)]
            outs[mode].append(sub.tap(feats, row_off, len(frames), R, max(frames), 101)[1].clone())
    for i in range(depth):
        a, b = outs["fused"][i].float(), outs["generic"][i].float()
        assert torch.isfinite(a).all()
        rel = ((a - b).norm() / b.norm()).item()
        print(name, "block", i, "res2 rel L2 fused vs per-step %.2e" % rel)
        assert rel < 1.5e-2, (name, i, rel)
    cos = F.cosine_similarity(outs["fused"][-1].double(), outs["generic"][-1].double(), dim=1)
    assert cos.min() > 0.9995


def test_wide_res2_time_segments_are_bit_identical(wide_supernet):
    from sugar_amd import _lib
    cfg = R2_WIDE_CASES["w128k3"]
    sub = wide_supernet.clone(cfg).set_precision("bf16")
    wide_supernet.config = None
    frames = [384, 301, 129, 16, 383, 200]
    R = sum(frames)
    feats = synth.normal("res2wideseg/feats", (R, 80), seed=10).to(torch.bfloat16).to(DEV)
    row_off = torch.tensor(_row_off(frames), dtype=torch.int32, device=DEV)
    outs = {}
    for cap in (384, 256, 160, 0):                    # 384: whole utterances; 0: the library's own few-utterance rule
        with _lib.tuned(res2_max_t=cap):
            outs[cap] = [sub.tap(feats, row_off, len(frames), R, max(frames), 2 + 4 * i)[0].clone() for i in range(cfg[0])]
    for cap in (256, 160, 0):
        for i in range(cfg[0]):
            assert torch.equal(outs[384][i], outs[cap][i]), (cap, i)
