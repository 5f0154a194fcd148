"""GPU: each HIP kernel, called through the C-ABI, against the oracle on the same seeded inputs."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sugar_amd
from conftest import cfg_of
from oracle import ref_cpu
from sugar_amd import _lib, layout, synth
from sugar_amd.transforms import BatchPlan

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _row_off(frames):
    off = [0]
    for t in frames:
        off.append(off[-1] + t)
    return off


def _conv_case(dtype, cin, cout, k, dil, flags, frames, with_x2, seed):
    """Run sugar_conv1d_fwd on packed channels-last rows and the oracle (F.conv1d per utterance) on the same data."""
    tdt = torch.float32 if dtype == _lib.SUGAR_F32 else torch.bfloat16
    R = sum(frames)
    off = _row_off(frames)
    x = synth.normal("conv/x", (R, cin), seed).to(tdt)
    x2 = synth.normal("conv/x2", (R, cin), seed).to(tdt) if with_x2 else None
    w = synth.uniform("conv/w", (cout, cin, k), -1.0, 1.0, seed) * (3.0 / (cin * k)) ** 0.5
    bn = synth.fill_state_dict({"b.weight": (cout,), "b.bias": (cout,), "b.running_mean": (cout,), "b.running_var": (cout,)}, seed)
    # oracle on the values the kernel actually sees (inputs / weights rounded to the compute dtype)
    wq = w.to(tdt).float()
    xin = x.float() + (x2.float() if with_x2 else 0)
    if with_x2:
        xin = xin.to(tdt).float()
    outs = []
    for b in range(len(frames)):
        seg = xin[off[b]:off[b + 1]].t().unsqueeze(0)
        y = F.conv1d(seg, wq, None, 1, ((k - 1) // 2) * dil, dil)
        if flags & 1:
            y = F.relu(y)
        if flags & 2:
            y = F.batch_norm(y, bn["b.running_mean"], bn["b.running_var"], bn["b.weight"], bn["b.bias"], False, 0.0, 1e-5)
        if flags & 4:
            y = torch.tanh(y)
        outs.append(y[0].t())
    want = torch.cat(outs)
    xd = x.to(DEV)
    x2d = x2.to(DEV) if with_x2 else None
    yd = torch.full((R, cout), float("nan"), dtype=tdt, device=DEV)
    rowd = torch.tensor(off, dtype=torch.int32, device=DEV)
    cbn = _lib.SugarBN(*[_lib.fptr(bn["b." + n]) for n in ("weight", "bias", "running_mean", "running_var")])
    wc = w.contiguous()
    rc = _lib.lib().sugar_conv1d_fwd(_lib.fptr(wc), C.byref(cbn), cout, cin, k, dil, flags, xd.data_ptr(),
                                     x2d.data_ptr() if with_x2 else None, yd.data_ptr(), rowd.data_ptr(), len(frames), R,
                                     max(frames), dtype, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "sugar_conv1d_fwd")
    torch.cuda.synchronize()
    return yd.float().cpu(), want


CONV_CASES = [
    # cin, cout, k, dil, flags, frames, x2
    (80, 512, 5, 1, 3, [63], False),                      # layer1 (K = 400 -> padded K tile); bf16: LDS-DMA GEMM over taps
    (80, 512, 5, 1, 3, [301, 17, 130, 63, 2, 1], False),  # ... ragged utterances: every zero-padded boundary of the tap GEMM
    (512, 512, 1, 1, 3, [301, 301], False),               # first/last 1x1 (weights-in-registers GEMM; most workgroups idle)
    (512, 512, 1, 1, 3, [301] * 40, False),               # ... 189 frame tiles: 1-2 per workgroup, the ring wraps
    (512, 512, 1, 1, 7, [1500] * 20 + [7], False),        # ... 3-4 tiles per workgroup, ragged last tile, tanh epilogue
    (512, 512, 1, 1, 3, [2048] * 25 + [1], False),        # ... R = 51201 >= the switch-over: the kernel the bench runs
    (384, 384, 1, 1, 3, [301] * 30 + [5], False),         # narrower subnets (128 x 128 tile kernel)
    (320, 448, 1, 1, 3, [301] * 30 + [5], False),
    (64, 64, 3, 2, 3, [301, 17, 130], True),              # Res2 step, ragged utterances, addend
    (64, 64, 5, 5, 3, [301, 63], True),                   # widest halo: k 5, dilation 5
    (16, 16, 1, 3, 3, [63, 63], True),                    # narrow subnet width (padded to the 64-row tile)
    (24, 24, 3, 4, 3, [40, 200], False),                  # width 22 padded to 24
    (1536, 1536, 1, 1, 1, [150], False),                  # catconv: ReLU only (6 tiles: all in the split tail)
    (1536, 1536, 1, 1, 1, [301] * 40, False),             # ... 288 tiles: one full round of 256 x 256 ping-pong tiles + tail
    (2048, 1536, 1, 1, 1, [301] * 40 + [9], False),       # ... Largest's catconv (K = 2048), ragged last tile
    (1536, 128, 1, 1, 4, [129], False),                   # pooling.linear1 + tanh
    (176, 48, 1, 1, 0, [5, 1, 300], False),               # odd sizes, no epilogue, 1-frame utterance
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "cin%d_cout%d_k%d_d%d_f%d" % c[:5])
def test_conv_f32(case):
    got, want = _conv_case(_lib.SUGAR_F32, *case[:6], with_x2=case[6], seed=1)
    assert torch.isfinite(got).all()
    assert _rel(got, want) < 2e-5


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "cin%d_cout%d_k%d_d%d_f%d" % c[:5])
def test_conv_bf16(case):
    got, want = _conv_case(_lib.SUGAR_BF16, *case[:6], with_x2=case[6], seed=2)
    assert torch.isfinite(got).all()
    # inputs/weights are identical bf16 values on both sides; the only difference is the bf16 rounding of the output
    assert _rel(got, want) < 6e-3


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[0] == 512 and c[1] == 512 and c[2] == 1],
                         ids=lambda c: "frames%dx%d_f%d" % (len(c[5]), c[5][0], c[4]))
def test_conv512_weights_in_registers_is_bit_identical_to_tiles(case):
    """The K = 512 convs switch to gemm_wreg_kernel at R >= 19200; both kernels must give the same bits at any R."""
    with _lib.tuned(gemm_wreg_min_r=0):
        wreg, want = _conv_case(_lib.SUGAR_BF16, *case[:6], with_x2=False, seed=5)
    with _lib.tuned(gemm_wreg_min_r=1 << 40):
        tiles, _ = _conv_case(_lib.SUGAR_BF16, *case[:6], with_x2=False, seed=5)
    assert torch.equal(wreg, tiles)
    assert _rel(wreg, want) < 6e-3


@pytest.mark.parametrize("frames", [[129], [301] * 40, [1500] * 9 + [7]], ids=["129", "40x301", "ragged"])
def test_pool1_weights_in_registers_is_bit_identical_to_tiles(frames):
    """pooling.linear1 (1536 -> 128, tanh) switches to gemm_wreg_kernel<24, 2, 2, 4> with the 512-channel convs."""
    case = (1536, 128, 1, 1, 4, frames)
    with _lib.tuned(gemm_wreg_min_r=0):
        wreg, want = _conv_case(_lib.SUGAR_BF16, *case, with_x2=False, seed=8)
    with _lib.tuned(gemm_wreg_min_r=1 << 40):
        tiles, _ = _conv_case(_lib.SUGAR_BF16, *case, with_x2=False, seed=8)
    assert torch.equal(wreg, tiles)
    assert _rel(wreg, want) < 6e-3


@pytest.mark.parametrize("cin,cout", [(1536, 1536), (1152, 1152)], ids=["1536", "1152"])
@pytest.mark.parametrize("frames", [[301] * 40, [301] * 100, [301] * 110 + [70]], ids=["1round", "2rounds+tail", "3rounds+split4"])
def test_wide_gemm_schedules_are_bit_identical(frames, cin, cout):
    """catconv shape with full 256 x 256 rounds.  What the forward runs -- split tail, and from two rounds up ONE
    persistent workgroup per CU on the quadrant schedule whose DMA stream runs across tile boundaries -- against the
    persistent two-half ping-pong, one workgroup per tile, the in-step schedule and the uniform grid: the same bytes,
    and F.conv1d within bf16.
    1152 channels: 4.5 channel tiles (the last one half empty: masked stores, uncounted vmcnt wait) and 18 k-tiles."""
    case = (cin, cout, 1, 1, 1, frames)
    pp, want = _conv_case(_lib.SUGAR_BF16, *case, with_x2=False, seed=6)
    with _lib.tuned(gemm_no_persist=1):
        per_tile, _ = _conv_case(_lib.SUGAR_BF16, *case, with_x2=False, seed=6)
    for sched in (1, 2):                             # persistent loop on the two-half ping-pong / the four-phase quadrant schedule
        with _lib.tuned(gemm_no_quad=sched):
            other, _ = _conv_case(_lib.SUGAR_BF16, *case, with_x2=False, seed=6)
        assert torch.equal(pp, other), sched
    with _lib.tuned(gemm_no_pingpong=1):
        instep, _ = _conv_case(_lib.SUGAR_BF16, *case, with_x2=False, seed=6)
    with _lib.tuned(gemm_uniform=1):
        uniform, _ = _conv_case(_lib.SUGAR_BF16, *case, with_x2=False, seed=6)
    assert torch.isfinite(pp).all()
    assert torch.equal(pp, per_tile) and torch.equal(pp, instep) and torch.equal(pp, uniform)
    assert _rel(pp, want) < 6e-3


@pytest.mark.parametrize("frames", [[63], [301, 17, 130, 63, 2, 1], [301] * 20 + [64, 1, 65]], ids=["one", "ragged", "20x301"])
def test_layer1_tap_gemm_and_generic_conv_agree_with_oracle(frames):
    """layer1 (80 -> 512, k 5) runs as an LDS-DMA GEMM over overlapping zero-padded rows -- 128 x 128 tiles with a
    per-row utterance search for small inputs, the weights-in-registers kernel over the padded row space with an
    output-row map for large ones; the generic masked-load conv is a third implementation of the same sums.  All three
    against F.conv1d; the two GEMM forms sum in the same order and must agree bit for bit."""
    case = (80, 512, 5, 1, 3, frames)
    with _lib.tuned(gemm_wreg_min_r=1 << 40):
        tiles, want = _conv_case(_lib.SUGAR_BF16, *case, with_x2=False, seed=7)
    with _lib.tuned(gemm_wreg_min_r=0):
        wreg, _ = _conv_case(_lib.SUGAR_BF16, *case, with_x2=False, seed=7)
    with _lib.tuned(generic_layer1=1):
        generic, _ = _conv_case(_lib.SUGAR_BF16, *case, with_x2=False, seed=7)
    assert _rel(tiles, want) < 6e-3 and _rel(generic, want) < 6e-3
    assert torch.isfinite(wreg).all() and torch.equal(wreg, tiles)
    # same products, different summation order (k-tile order vs tap order): only f32 reassociation + one bf16 ulp
    assert _rel(tiles, generic) < 4e-3


def test_conv_rejects_bad_arguments():
    x = torch.zeros(8, 12, device=DEV)
    w = torch.zeros(8, 12, 1)
    off = torch.tensor([0, 8], dtype=torch.int32, device=DEV)
    rc = _lib.lib().sugar_conv1d_fwd(_lib.fptr(w), None, 8, 12, 1, 1, 0, x.data_ptr(), None, x.data_ptr(), off.data_ptr(),
                                     1, 8, 8, _lib.SUGAR_F32, None)
    assert rc == -1
    rc = _lib.lib().sugar_conv1d_fwd(_lib.fptr(w), None, 8, 16, 4, 1, 0, x.data_ptr(), None, x.data_ptr(), off.data_ptr(),
                                     1, 8, 8, _lib.SUGAR_F32, None)
    assert rc == -1 and b"bad shape" in _lib.lib().sugar_last_error()


# ------------------------------------------------------------------------------------------ front-end
# Absolute error on instance-normalised (unit-variance) log-mel features.  Two oracles: the f32 restatement (torch.stft)
# and the independent float64 one (numpy framing + rFFT); they differ from each other by up to 2.3e-4 on 3 s utterances
# (outliers in the three lowest mel bins, which hold one or two low-energy FFT bins each: the log amplifies the f32
# rounding of |X|^2), 6e-5 at the 99.99th percentile.  The HIP front-end (f32 radix-4 FFT) shows the same kind of outlier.
# Measured on MI355X (printed by the tests), max over all values: vs the float64 oracle 2.8e-5 (10 000 samples), 1.1e-4
# (5 x 3 s), 3.4e-5 .. 3.6e-4 (ragged list, worst on its 3 s entry); vs the f32 oracle 5.9e-5 / 5.4e-4 / 6e-5 .. 3.5e-4.
# Bounds: 1e-3 against float64 (< 3x the worst measured), 1.5e-3 against the f32 oracle (< 3x).  Round 1 allowed 2e-3
# against the f32 oracle only and did not record the measurement.
FBANK_TOL = 1e-3            # vs the float64 oracle
FBANK_TOL_F32 = 1.5e-3      # vs the f32 oracle (its own outliers included)


def _fbank_f64(wav_1d):
    return torch.from_numpy(ref_cpu.logmel_fbank_np64(wav_1d.numpy())).float()      # [80, T]


def test_fbank_cfg1_matches_oracle(golden):
    wav = synth.waveforms(1, 10000, seed=0)
    got = sugar_amd.LogMelFbanks()(wav.to(DEV)).cpu()
    assert got.shape == (1, 1, 80, 63)
    want = torch.from_numpy(golden["fbank/cfg1"])
    e1 = (got[0, 0] - want).abs().max().item()
    e2 = (got[0, 0] - ref_cpu.logmel_fbank(wav)[0, 0]).abs().max().item()
    e3 = (got[0, 0] - _fbank_f64(wav[0])).abs().max().item()
    print("fbank cfg1: max abs err vs golden %.2e, vs f32 oracle %.2e, vs f64 oracle %.2e" % (e1, e2, e3))
    assert e1 < FBANK_TOL_F32 and e2 < FBANK_TOL_F32 and e3 < FBANK_TOL


def test_fbank_batch_and_ragged_match_oracle():
    fb = sugar_amd.LogMelFbanks()
    wav = synth.waveforms(5, 48000, seed=4)
    got = fb(wav.to(DEV)).cpu()
    want = ref_cpu.logmel_fbank(wav)
    assert got.shape == want.shape == (5, 1, 80, 301)
    e32 = (got - want).abs().max().item()
    e64 = max((got[i, 0] - _fbank_f64(wav[i])).abs().max().item() for i in range(5))
    print("fbank 5 x 3 s: max abs err vs f32 oracle %.2e, vs f64 oracle %.2e" % (e32, e64))
    assert e32 < FBANK_TOL_F32 and e64 < FBANK_TOL
    # ragged: lengths straddling the 64-frame tile and the 160-sample hop; each utterance normalised on its own frames
    lens = [257, 10239, 10240, 10241, 48000, 20479, 66666]
    wavs = synth.ragged_waveforms(lens, seed=9)
    plan = BatchPlan(lens, DEV)
    rows = fb.rows(torch.cat(wavs).to(DEV), plan).cpu()
    off = plan.row_off.cpu().tolist()
    for i, wv in enumerate(wavs):
        want = ref_cpu.logmel_fbank(wv.unsqueeze(0))[0, 0].t()
        assert rows[off[i]:off[i + 1]].shape == want.shape
        e32 = (rows[off[i]:off[i + 1]] - want).abs().max().item()
        e64 = (rows[off[i]:off[i + 1]] - _fbank_f64(wv).t()).abs().max().item()
        print("fbank ragged: %6d samples, max abs err vs f32 oracle %.2e, vs f64 oracle %.2e" % (lens[i], e32, e64))
        assert e32 < FBANK_TOL_F32 and e64 < FBANK_TOL, (lens[i], e32, e64)
    # bf16 output is the rounded f32 output
    rows16 = fb.rows(torch.cat(wavs).to(DEV), plan, torch.bfloat16).float().cpu()
    assert (rows16 - rows).abs().max() < 0.04


# ------------------------------------------------------------------------------------------ backbone stages
def _unpack(act, name, dims, i=None):
    """Internal padded channel layout -> reference channel order."""
    if name in ("first", "res2"):
        w, wp = dims["w"][i], dims["wp"][i]
        idx = torch.tensor([g * wp + j for g in range(8) for j in range(w)])
        return act[:, idx]
    return act


@pytest.mark.parametrize("name", ["smallest", "odd"])
def test_stage_taps_match_reference_f32(name, golden, meta, supernet):
    cfg = cfg_of(meta, name)
    sub = supernet.clone(cfg).set_precision("fp32")
    supernet.config = None
    x = ref_cpu.logmel_fbank(synth.waveforms(1, 10000, seed=0)).squeeze(1)      # oracle features -> backbone only
    feats = x[0].t().contiguous().to(DEV)
    row_off = torch.tensor([0, 63], dtype=torch.int32, device=DEV)
    dims = sub.packed_dims()
    depth, ch, _, ccat = cfg
    taps = [(0, "layer1", None)]
    for i in range(depth):
        taps += [(1 + 4 * i, "b%d.first" % i, i), (2 + 4 * i, "b%d.res2" % i, i), (3 + 4 * i, "b%d.last" % i, i),
                 (4 + 4 * i, "b%d.out" % i, i)]
    taps += [(100, "catconv", None)]
    errs = {}
    for what, key, i in taps:
        act, _ = sub.tap(feats, row_off, 1, 63, 63, what)
        act = _unpack(act.cpu(), key.split(".")[-1], dims, i)
        want = torch.from_numpy(golden["stage/%s/%s" % (name, key)]).t()
        errs[key] = _rel(act[:, :want.shape[1]], want)
    pooled, emb = sub.tap(feats, row_off, 1, 63, 63, 101)
    pooled = pooled.cpu()[0]
    cp = dims["ccatp"]
    want = torch.from_numpy(golden["stage/%s/pool" % name])
    got = torch.cat([pooled[:ccat], pooled[cp:cp + ccat]])
    errs["pool"] = _rel(got, want)
    errs["embed"] = _rel(emb.cpu()[0], torch.from_numpy(golden["stage/%s/embed" % name]))
    print(name, {k: "%.2e" % v for k, v in errs.items()})
    for k, v in errs.items():
        assert v < 1e-4, (k, errs)


# bf16 path, stage by stage, against the same reference module outputs.  Every stage reads bf16-rounded inputs (relative
# rounding 2^-9 = 2e-3 per element) and the error random-walks through ~7 tensors per block, so the bound is on the
# relative L2 error of the whole activation: 1.5e-2 = 3x the largest measured value (MI355X, printed: 2e-3 .. 3e-3 at
# layer1 / first_conv, growing to 4.2e-3 ("smallest") and 5.0e-3 ("odd") at catconv; pooled statistics 1.6e-3).
# This is the per-stage check of the fast bf16 kernels (tap GEMM layer1, LDS-DMA 1x1 GEMMs, fused Res2 chain, the
# epilogue SE sums) that the f32 test above cannot give: they only exist in bf16.
@pytest.mark.parametrize("name", ["smallest", "odd"])
def test_stage_taps_match_reference_bf16(name, golden, meta, supernet):
    cfg = cfg_of(meta, name)
    sub = supernet.clone(cfg).set_precision("bf16")
    supernet.config = None
    x = ref_cpu.logmel_fbank(synth.waveforms(1, 10000, seed=0)).squeeze(1)
    feats = x[0].t().contiguous().to(torch.bfloat16).to(DEV)
    row_off = torch.tensor([0, 63], dtype=torch.int32, device=DEV)
    dims = sub.packed_dims()
    depth, ch, _, ccat = cfg
    taps = [(0, "layer1", None)]
    for i in range(depth):
        taps += [(1 + 4 * i, "b%d.first" % i, i), (2 + 4 * i, "b%d.res2" % i, i), (3 + 4 * i, "b%d.last" % i, i),
                 (4 + 4 * i, "b%d.out" % i, i)]
    taps += [(100, "catconv", None)]
    errs = {}
    for what, key, i in taps:
        act, _ = sub.tap(feats, row_off, 1, 63, 63, what)
        act = _unpack(act.float().cpu(), key.split(".")[-1], dims, i)
        want = torch.from_numpy(golden["stage/%s/%s" % (name, key)]).t()
        errs[key] = ((act[:, :want.shape[1]].double() - want.double()).norm() / want.double().norm()).item()
    pooled, emb = sub.tap(feats, row_off, 1, 63, 63, 101)
    cp = dims["ccatp"]
    pooled = pooled.cpu()[0]
    want = torch.from_numpy(golden["stage/%s/pool" % name])
    got = torch.cat([pooled[:ccat], pooled[cp:cp + ccat]])
    errs["pool"] = ((got.double() - want.double()).norm() / want.double().norm()).item()
    want = torch.from_numpy(golden["stage/%s/embed" % name])
    errs["embed"] = ((emb.cpu()[0].double() - want.double()).norm() / want.double().norm()).item()
    print(name, "bf16 rel L2 per stage", {k: "%.2e" % v for k, v in errs.items()})
    for k, v in errs.items():
        assert v < 1.5e-2, (k, errs)


# Full-width networks at T = 301, stage by stage, against the REFERENCE's module outputs (tests/golden/round3.npz,
# oracle/make_golden_round3.py; every tap stored at the 24 frames of oracle/round3_inputs.py::FRAMES, all channels).
# This is the oracle check of the kernels only full-width subnets reach: res2_chain_kernel on 64-channel groups of 3
# taps, res2_wide_kernel on 128-channel groups of 5 taps, the weights-in-registers 512 x 512 GEMM, the 1024-channel
# GEMMs, the 256 x 256 catconv.  f32: same 1e-4 bar as the narrow subnets; bf16: relative L2 over the stored rows
# < 1.5e-2 (same bar and reasoning as test_stage_taps_match_reference_bf16; measured values are printed).
# "bf16-wreg": the same with gemm_wreg_min_r = 0, i.e. every K = 512 conv, layer1 and pooling.linear1 on the
# weights-in-registers GEMM that large batches use (one utterance alone stays below its threshold).
@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16-wreg"])
@pytest.mark.parametrize("net", ["base", "wide_full"])
def test_stage_taps_full_width_T301_match_reference(net, precision):
    if precision == "bf16-wreg":
        with _lib.tuned(gemm_wreg_min_r=0):
            return _stage_taps_T301(net, "bf16")
    return _stage_taps_T301(net, precision)


def _stage_taps_T301(net, precision):
    import os
    from oracle.round3_inputs import FRAMES, NETS
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round3.npz"))
    family, shape_args, cfg = NETS[net]
    sup = getattr(sugar_amd, family)()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(*shape_args), seed=0))
    sub = sup.clone(cfg).set_precision(precision)
    x = ref_cpu.logmel_fbank(synth.waveforms(1, 48000, seed=5)).squeeze(1)      # oracle features -> backbone only
    T = x.shape[2]
    assert T == 301
    feats = x[0].t().contiguous().to(torch.bfloat16 if precision == "bf16" else torch.float32).to(DEV)
    row_off = torch.tensor([0, T], dtype=torch.int32, device=DEV)
    dims = sub.packed_dims()
    depth, ch, _, ccat = cfg
    taps = [(0, "layer1", None)]
    for i in range(depth):
        taps += [(1 + 4 * i, "b%d.first" % i, i), (2 + 4 * i, "b%d.res2" % i, i), (3 + 4 * i, "b%d.last" % i, i),
                 (4 + 4 * i, "b%d.out" % i, i)]
    taps += [(100, "catconv", None)]
    errs = {}
    rel = lambda got, want: ((got.double() - want.double()).norm() / want.double().norm()).item()
    for what, key, i in taps:
        act, _ = sub.tap(feats, row_off, 1, T, T, what)
        act = _unpack(act.float().cpu(), key.split(".")[-1], dims, i)[FRAMES]
        want = torch.from_numpy(gold["stage301/%s/%s" % (net, key)]).t()          # [24 frames, C]
        errs[key] = rel(act[:, :want.shape[1]], want)
    pooled, emb = sub.tap(feats, row_off, 1, T, T, 101)
    pooled, cp = pooled.cpu()[0], dims["ccatp"]
    errs["pool"] = rel(torch.cat([pooled[:ccat], pooled[cp:cp + ccat]]), torch.from_numpy(gold["stage301/%s/pool" % net]))
    errs["embed"] = rel(emb.cpu()[0], torch.from_numpy(gold["stage301/%s/embed" % net]))
    print(net, precision, "rel L2 per stage at T = 301:", {k: "%.2e" % v for k, v in errs.items()})
    bar = 1e-4 if precision == "fp32" else 1.5e-2
    for k, v in errs.items():
        assert v < bar, (k, errs)
    sub.release()


# ------------------------------------------------------------------------------------------ scoring / l2norm
def test_cosine_trials_match_calculate_score(golden, meta):
    segs = meta["score_segs"]
    vecs = [synth.normal("score/vec%d" % i, (segs[i], 192), seed=2) for i in range(len(segs))]
    vecs = [F.normalize(v, p=2, dim=1) if i % 2 == 0 else v for i, v in enumerate(vecs)]
    lab, a, b = synth.trial_pairs(len(segs), 300, seed=2)
    vectors = {"utt%03d" % i: v.to(DEV) for i, v in enumerate(vecs)}
    trials = [{"label": int(l), "enroll": "utt%03d" % i, "test": "utt%03d" % j} for l, i, j in zip(lab, a, b)]
    scores, labels, pairs = sugar_amd.score_trials(trials, vectors, mode="cosine", n_jobs=1)
    want = golden["score/scores"]
    assert scores.dtype == np.float64 and scores.shape == want.shape
    assert np.abs(scores - want).max() < 2e-6
    assert np.array_equal(labels, lab) and pairs[0] == [trials[0]["enroll"], trials[0]["test"]]
    one = sugar_amd.calculate_score(vecs[3].to(DEV), vecs[8].to(DEV), "cosine")
    assert abs(one - ref_cpu.cosine_score(vecs[3], vecs[8])) < 2e-6


def test_cosine_trials_full_size_config4_against_reference():
    """BASELINE config 4 at full size: 4 874 unit vectors, the 37 720-pair list, against the reference's own
    calculate_score outputs (tests/golden/round2.npz).  Random unit vectors in 192-d give 37 720 scores inside +-0.35, so
    27 213 neighbouring gaps are below 1e-5 (some pairs are drawn twice: gap 0): "identical ordering" is asserted as (a)
    identical argsort on the near-tie-free sub-list and (b) no inversion larger than the arithmetic error anywhere.
    Also times the kernel (trials/s, achieved GB/s over N*192*4 + trials*12 bytes) into gpurun_out/score_full.log."""
    import os
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = np.load(os.path.join(root, "tests", "golden", "round2.npz"))["score/full"]
    n_utts, n_trials = 4874, 37720
    vec = F.normalize(synth.normal("score4/vec", (n_utts, 192), seed=4), p=2, dim=1).to(DEV)
    lab, a, b = synth.trial_pairs(n_utts, n_trials, seed=4)
    seg = torch.arange(n_utts + 1, dtype=torch.int32, device=DEV)
    e = torch.from_numpy(a.astype(np.int32)).to(DEV)
    t = torch.from_numpy(b.astype(np.int32)).to(DEV)
    from sugar_amd.scores import score_pairs
    got = score_pairs(vec, seg, e, t).double().cpu().numpy()
    err = np.abs(got - want).max()
    assert err < 2e-6, err
    order = np.argsort(want, kind="stable")
    keep = [order[0]]
    for i in order[1:]:
        if want[i] - want[keep[-1]] > 1e-5:
            keep.append(i)
    keep = np.array(sorted(keep))
    assert len(keep) > 5000
    assert np.array_equal(np.argsort(got[keep], kind="stable"), np.argsort(want[keep], kind="stable"))
    by_gpu = want[np.argsort(got, kind="stable")]                     # reference scores in the GPU's order
    assert (by_gpu[:-1] - by_gpu[1:]).max() < 4e-6                   # never decreasing by more than twice the error
    for _ in range(5):
        score_pairs(vec, seg, e, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        score_pairs(vec, seg, e, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    nbytes = n_utts * 192 * 4 + n_trials * 12
    line = ("cosine_trials_kernel, %d trials over %d vectors: %.1f us per call (host-paced), %.2e trials/s, %.1f GB/s of "
            "%d algorithmic bytes; max |score - reference| %.2e" % (n_trials, n_utts, dt * 1e6, n_trials / dt,
                                                                    nbytes / dt / 1e9, nbytes, err))
    print(line)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "score_full.log"), "w") as f:
        f.write(line + "\n")


def test_cosine_tiny_norm_uses_eps_per_vector():
    a = torch.tensor([[3e-9, 4e-9] + [0.0] * 190])
    b = torch.tensor([[1.0, 2.0, 2.0] + [0.0] * 189])
    got = sugar_amd.calculate_score(a.to(DEV), b.to(DEV), "cosine")
    assert abs(got - ref_cpu.cosine_score(a, b)) < 1e-6


def test_norm2_and_list_of_modes_match_reference():
    """calculate_score(mode="norm2") -- the reference's default -- and list-of-modes score_trials
    (sugar/scores.py:141-145, 151-152, 175-177) against the reference's own outputs (tests/golden/norm2.npz,
    oracle/make_golden_norm2.py): 40 utterances of 1-3 segment vectors, 300 trials.  Float bar: 2e-6 absolute on scores
    of magnitude 1-3 (f32 arithmetic on both sides, different summation order)."""
    import os
    from oracle.make_golden_norm2_inputs import inputs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = np.load(os.path.join(root, "tests", "golden", "norm2.npz"))
    vecs, lab, a, b = inputs()
    vectors = {"utt%03d" % i: v.to(DEV) for i, v in enumerate(vecs)}
    trials = [{"label": int(l), "enroll": "utt%03d" % i, "test": "utt%03d" % j} for l, i, j in zip(lab, a, b)]
    scores, labels, pairs = sugar_amd.score_trials(trials, vectors, mode="norm2", n_jobs=1)
    assert scores.shape == (300,) and np.abs(scores - gold["norm2"]).max() < 2e-6
    both, labels2, _ = sugar_amd.score_trials(trials, vectors, mode=["cosine", "norm2"], n_jobs=1)
    assert both.shape == (300, 2) and np.abs(both - gold["cosine_norm2"]).max() < 2e-6
    assert np.array_equal(labels, gold["labels"]) and np.array_equal(labels2, gold["labels"])
    assert np.array_equal(np.argsort(scores, kind="stable"), np.argsort(gold["norm2"], kind="stable"))
    # the default mode of calculate_score is norm2, as in the reference; a list gives a list
    assert abs(sugar_amd.calculate_score(vectors["utt001"], vectors["utt002"]) - float(gold["default_1_2"])) < 2e-6
    pair = sugar_amd.calculate_score(vectors["utt001"], vectors["utt002"], ["cosine", "norm2"])
    assert isinstance(pair, list) and abs(pair[1] - float(gold["default_1_2"])) < 2e-6
    assert abs(pair[0] - ref_cpu.cosine_score(vecs[1], vecs[2])) < 2e-6
    with pytest.raises(ValueError):
        sugar_amd.score_trials(trials, vectors, mode="plda")


def test_l2norm_matches_f_normalize():
    x = synth.normal("l2/x", (33, 192), seed=1)
    x[5] = 0
    xd = x.to(DEV).clone()
    _lib.check(_lib.lib().sugar_l2norm(xd.data_ptr(), 33, 192, torch.cuda.current_stream().cuda_stream))
    assert (xd.cpu() - F.normalize(x, p=2, dim=1)).abs().max() < 1e-6


# ------------------------------------------------------------------------------------------ fused Res2 chain
@pytest.mark.parametrize("name", ["base", "largest", "smallest", "odd", "rand0"])
def test_fused_res2_chain_matches_per_step_path_bf16(name, meta, supernet):
    """The one-kernel Res2 chain (res2.hip) against the 7-launch generic path, same bf16 inputs, ragged utterances."""
    cfg = cfg_of(meta, name)
    sub = supernet.clone(cfg).set_precision("bf16")
    supernet.config = None
    # <= 384 frames: one workgroup per utterance; "largest"/"odd" also run time-segmented utterances
    frames = [301, 63, 17, 130] if name not in ("largest", "odd") else [640, 385, 16, 1500]
    R = sum(frames)
    feats = synth.normal("res2/feats", (R, 80), seed=4).to(torch.bfloat16).to(DEV)
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
        # the fused chain adds x_j + sp_{j-1} before one bf16 rounding (the per-step path rounds sp first): the two
        # differ by bf16 rounding noise propagated through <= 7 steps and i blocks, nothing larger
        rel = ((a - b).norm() / b.norm()).item()
        print(name, "block", i, "res2 rel L2 fused vs per-step %.2e" % rel)
        assert rel < 1.5e-2, (name, i, rel)
    cos = F.cosine_similarity(outs["fused"][-1].double(), outs["generic"][-1].double(), dim=1)
    assert cos.min() > 0.9995


@pytest.mark.parametrize("name", ["base", "largest", "odd"])
def test_res2_time_segments_are_bit_identical(name, meta, supernet):
    """Utterances cut into time segments with 7 * halo context frames (res2.hip) give exactly the bytes of the chain
    run over the whole utterance in one workgroup: SUGAR_RES2_MAX_T lowers the per-workgroup capacity so that
    utterances the kernel can hold whole (<= 384 frames) are segmented, and both runs use the same fused kernel."""
    cfg = cfg_of(meta, name)
    sub = supernet.clone(cfg).set_precision("bf16")
    supernet.config = None
    frames = [384, 301, 129, 16, 383, 200]
    R = sum(frames)
    feats = synth.normal("res2seg/feats", (R, 80), seed=5).to(torch.bfloat16).to(DEV)
    row_off = torch.tensor(_row_off(frames), dtype=torch.int32, device=DEV)
    depth = cfg[0]
    outs = {}
    for cap in (None, "256", "160", "auto"):          # None: whole utterances (384 = the kernel's capacity); auto: the
        with _lib.tuned(res2_max_t={None: 384, "auto": 0}.get(cap) if cap in (None, "auto") else int(cap)):   # few-utterance rule
            outs[cap] = [sub.tap(feats, row_off, len(frames), R, max(frames), 2 + 4 * i)[0].clone() for i in range(depth)]
            outs[cap].append(sub.tap(feats, row_off, len(frames), R, max(frames), 101)[1].clone())
    # the kernel without the double-buffered u tile (two barriers per step, staging tiles of its own): same bytes
    for cap, v in (("whole/no_db", 384), ("160/no_db", 160)):
        with _lib.tuned(res2_max_t=v, res2_no_db=1):
            outs[cap] = [sub.tap(feats, row_off, len(frames), R, max(frames), 2 + 4 * i)[0].clone() for i in range(depth)]
            outs[cap].append(sub.tap(feats, row_off, len(frames), R, max(frames), 101)[1].clone())
    for cap in ("256", "160", "auto", "whole/no_db", "160/no_db"):
        for i in range(depth + 1):
            assert torch.equal(outs[None][i], outs[cap][i]), (name, cap, i)
