"""
GPU: end-to-end parity of the HIP path (through the C-ABI) with the reference.

Tolerances are BASELINE.json's: per-embedding cosine >= 0.9999 for the f32 path and >= 0.999 for the
bf16-operand / f32-accumulate path, against golden embeddings produced by the real reference backbone
(tests/golden, oracle/make_golden.py); identical trial ordering for scoring.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sugar_amd
from conftest import cfg_of
from oracle import ref_cpu
from sugar_amd import _lib, layout, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {"fp32": 0.9999, "bf16": 0.999}
# White-noise utterances give nearly parallel embeddings (pairwise cosine ~0.99), so the cosine bar alone is weak:
# the relative L2 error of each embedding is bounded as well.
REL = {"fp32": 1e-4, "bf16": 3e-2}


def _relerr(a, b):
    a, b = a.double().cpu(), torch.as_tensor(b).double()
    return ((a - b).norm(dim=1) / b.norm(dim=1)).max().item()


def _cos(a, b):
    return F.cosine_similarity(a.double().cpu(), torch.as_tensor(b).double(), dim=1)


def _model(supernet, cfg, precision):
    sub = supernet.clone(cfg).set_precision(precision)
    supernet.config = None
    return sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config1_base_one_10000_sample_waveform(precision, golden, supernet):
    model = _model(supernet, layout.BASE, precision)
    emb = model(synth.waveforms(1, 10000, seed=0).to(DEV))
    assert emb.shape == (1, 192) and emb.dtype == torch.float32
    c = _cos(emb, golden["embed/base/cfg1"])
    print("cfg1", precision, "cosine", c.tolist())
    assert c.min() >= TOL[precision]
    if precision == "fp32":
        assert (emb.cpu() - torch.from_numpy(golden["embed/base/cfg1"])).abs().max() < 5e-3


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["base", "largest", "rand0", "rand1", "rand2", "odd"])
def test_3s_batches_all_subnets(name, precision, golden, meta, supernet):
    model = _model(supernet, cfg_of(meta, name), precision)
    emb = model(synth.waveforms(4, 48000, seed=0).to(DEV))
    c = _cos(emb, golden["embed/%s/4x3s" % name])
    r = _relerr(emb, golden["embed/%s/4x3s" % name])
    print(name, precision, "cosine min %.6f  rel L2 err max %.2e" % (c.min(), r))
    assert c.min() >= TOL[precision]
    assert r < REL[precision]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_variable_length_equals_reference_one_at_a_time(precision, golden, meta, supernet):
    """BASELINE config 4: packed ragged batch == the reference run per utterance, unpadded (its batch_size=1 mode)."""
    model = _model(supernet, layout.BASE, precision)
    wavs = [w.to(DEV) for w in synth.ragged_waveforms(meta["ragged_lengths"], seed=7)]
    emb = model(wavs)
    c = _cos(emb, golden["embed/base/ragged"])
    r = _relerr(emb, golden["embed/base/ragged"])
    print("ragged", precision, c.tolist(), "rel L2 err max %.2e" % r)
    assert c.min() >= TOL[precision]
    assert r < REL[precision]
    # packing must not leak across utterance boundaries: each utterance alone gives the same embedding
    solo = torch.cat([model([w]) for w in wavs])
    assert (solo - emb).abs().max() < (1e-4 if precision == "fp32" else 1e-2)
    # order independence of the packed batch
    perm = [2, 0, 3, 1]
    emb_p = model([wavs[i] for i in perm])
    assert (emb_p - emb[perm]).abs().max() < (1e-4 if precision == "fp32" else 1e-2)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["base", "largest"])
def test_long_utterances_time_segments(name, precision, supernet):
    """Whole-utterance evaluation (the reference's num_samples=0 mode, sugar/waveforms.py:105-113) of 6.9 / 11 / 21 /
    1 s audio against the reference's own embeddings (tests/golden/long.npz): these lengths run the time-segmented
    Res2 chain (> 384 frames), the chunked instance norm (> 320) and the segmented attentive pooling (> 1024)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "long.npz"))
    cfg = layout.BASE if name == "base" else layout.LARGEST
    model = _model(supernet, cfg, precision)
    wavs = [w.to(DEV) for w in synth.ragged_waveforms([int(n) for n in g["lengths"]], seed=int(g["seed"]))]
    emb = model(wavs)
    want = g["embed/%s/long" % name]
    c, r = _cos(emb, want), _relerr(emb, want)
    print("long", name, precision, c.tolist(), "rel L2 err max %.2e" % r)
    assert c.min() >= TOL[precision]
    assert r < REL[precision]
    # an utterance's embedding does not depend on what else is in the batch (fixed segment lengths)
    solo = torch.cat([model([w]) for w in wavs])
    assert (solo - emb).abs().max() < (1e-4 if precision == "fp32" else 1e-2)
    if precision == "bf16":
        assert torch.equal(model([wavs[2], wavs[3]])[0], model([wavs[2]])[0])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_segment_boundary_lengths_match_oracle(precision, supernet, sup_state):
    """Utterances of exactly 3, 320/321 (instance-norm chunk), 384/385 (Res2 segment), 1024/1025 (pooling segment) and
    2048/2049 (SE segment) frames in ONE ragged batch against the CPU oracle run per utterance (the oracle itself is
    pinned to the reference at 101-2126 frames by tests/golden/long.npz)."""
    frames = [3, 320, 321, 384, 385, 1024, 1025, 2048, 2049]
    lens = [(t - 1) * 160 + 37 for t in frames]
    wavs = synth.ragged_waveforms(lens, seed=31)
    model = _model(supernet, layout.BASE, precision)
    emb = model([w.to(DEV) for w in wavs])
    want = ref_cpu.embed_ragged(ref_cpu.clone_state(sup_state, layout.BASE), layout.BASE[0], wavs)
    c, r = _cos(emb, want), _relerr(emb, want)
    print("boundary", precision, "cos min %.6f rel L2 max %.2e" % (c.min(), r))
    assert c.min() >= TOL[precision]
    assert r < REL[precision]


def test_supernet_with_active_config_equals_clone(golden, meta, supernet):
    """Reference property: sup.config = cfg; sup(x) == sup.clone(cfg)(x)."""
    cfg = cfg_of(meta, "rand0")
    x = ref_cpu.logmel_fbank(synth.waveforms(2, 10000, seed=0)).squeeze(1).to(DEV)
    supernet.precision = "fp32"
    supernet.config = cfg
    a = supernet.to(DEV)(x)
    b = supernet.clone(cfg).set_precision("fp32")(x)
    supernet.config = None
    supernet.precision = "bf16"
    assert torch.equal(a, b)
    want = ref_cpu.subnet_forward(ref_cpu.clone_state(supernet.state_dict(), cfg), cfg[0], x.cpu())
    assert _cos(a, want).min() >= 0.9999


def test_from_pretrained_files_end_to_end(tmp_path, golden, meta, sup_state):
    ck = {"module.__S__." + k: v for k, v in sup_state.items()}
    ck["module.__L__.W"] = torch.zeros(192, 10)
    torch.save({"state_dict": ck}, tmp_path / "depth.torchparams")
    shapes = layout.subnet_shapes(layout.BASE)
    bn = synth.fill_state_dict({k: shapes[k] for k in meta["bn_file_keys"]}, seed=11)
    torch.save({"subnet": layout.BASE, "bn": bn}, tmp_path / "base.bn.tar")
    subnet, _ = sugar_amd.WrappedModel.from_pretrained(str(tmp_path), "depth.torchparams", "base.bn.tar", precision="fp32")
    subnet = subnet.to(DEV).eval()
    emb = subnet(synth.waveforms(1, 10000, seed=0).to(DEV))         # README usage: embedding = subnet(wav)
    assert _cos(emb, golden["embed/base/cfg1_bnfile"]).min() >= 0.9999


class _Utts:
    """Stand-in for sugar.database.Utterance: items are {"file", "input": [n_seg, L]}."""

    def __init__(self, wavs, segs):
        self.wavs, self.segs = wavs, segs

    def __len__(self):
        return len(self.wavs)

    def __getitem__(self, i):
        w = self.wavs[i]
        n = self.segs[i]
        L = w.numel() // n
        return {"file": "utt%04d.wav" % i, "input": w[:n * L].view(n, L) if n > 1 else w.unsqueeze(0)}


def test_extract_vectors_and_score_trials_ordering(supernet):
    """extract_vectors dict layout + batched scoring: same score ORDER as the reference loop on the same vectors."""
    model = _model(supernet, layout.SMALLEST, "bf16")
    n_utts = 24
    lens = synth.ragged_lengths(n_utts, lo=32000, hi=80000, seed=3).tolist()
    segs = [1 + (i % 2) for i in range(n_utts)]
    data = _Utts(synth.ragged_waveforms(lens, seed=3), segs)
    vectors = sugar_amd.extract_vectors(model, data, norm=True, device=DEV, n_jobs=1, batch_size=8)
    assert list(vectors.keys()) == ["utt%04d.wav" % i for i in range(n_utts)]
    for i in range(n_utts):
        v = vectors["utt%04d.wav" % i]
        assert v.shape == (segs[i], 192)
        assert (v.norm(dim=1) - 1).abs().max() < 1e-5
    # batch composition must not matter.  batch_size=1 is the reference's whole-utterance mode (it runs those one at a
    # time); here the packer then chooses, and everything of this list goes out in ONE call
    calls = []
    orig = sugar_amd.SpeakerModel._embed_flat
    sugar_amd.SpeakerModel._embed_flat = lambda self, sub, flat, plan, nrm: (calls.append(plan.B), orig(self, sub, flat, plan, nrm))[1]
    try:
        v1 = sugar_amd.extract_vectors(model, data, norm=True, device=DEV, batch_size=1)
    finally:
        sugar_amd.SpeakerModel._embed_flat = orig
    assert calls == [sum(segs)]
    for k in vectors:
        assert (v1[k] - vectors[k]).abs().max() < 2e-2
    lab, a, b = synth.trial_pairs(n_utts, 400, seed=3)
    cpu_vecs = [vectors["utt%04d.wav" % i].cpu() for i in range(n_utts)]
    want = ref_cpu.score_trials(a, b, cpu_vecs)
    # drop near-ties (the ordering claim is for a list with gaps >> 1e-6, SURVEY section 7 "hard parts")
    order = np.argsort(want)
    keep = [order[0]]
    for i in order[1:]:
        if want[i] - want[keep[-1]] > 1e-5:
            keep.append(i)
    keep = np.array(sorted(keep))
    trials = [{"label": int(lab[i]), "enroll": "utt%04d.wav" % a[i], "test": "utt%04d.wav" % b[i]} for i in keep]
    scores, labels, _ = sugar_amd.score_trials(trials, vectors)
    assert np.abs(scores - want[keep]).max() < 2e-6
    assert np.array_equal(np.argsort(scores, kind="stable"), np.argsort(want[keep], kind="stable"))


class _Trials(list):
    """Stands in for sugar.database.VerificationTrials: iterable of trials + `.utterances`."""
    utterances = None


class _Loader:
    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size, self.num_workers = dataset, batch_size, 0


def test_veri_validate_matches_oracle_pipeline(supernet):
    """veri_validate (models/__init__.py:535-595): extraction + scoring on the GPU, EER / minDCF on the host; the same
    numbers as oracle embeddings + oracle scoring + the metrics module (itself pinned to the reference's metrics)."""
    from sugar_amd import metrics
    model = _model(supernet, layout.SMALLEST, "fp32")
    n_utts = 16
    lens = synth.ragged_lengths(n_utts, lo=24000, hi=48000, seed=9).tolist()
    wavs = synth.ragged_waveforms(lens, seed=9)
    lab, a, b = synth.trial_pairs(n_utts, 200, seed=9)
    trials = _Trials({"label": int(lab[i]), "enroll": "utt%04d.wav" % a[i], "test": "utt%04d.wav" % b[i]}
                     for i in range(len(lab)))
    trials.utterances = _Utts(wavs, [1] * n_utts)
    eer, dcf, vectors, scs = sugar_amd.veri_validate(_Loader(trials, 4), model, device=DEV, p_target=0.05,
                                                     ret_info=True)
    osd = ref_cpu.clone_state(synth.fill_state_dict(layout.supernet_shapes(), seed=0), layout.SMALLEST)
    want_vecs = [ref_cpu.embed(osd, layout.SMALLEST[0], w.unsqueeze(0), norm=True) for w in wavs]
    want_scs = ref_cpu.score_trials(a, b, want_vecs)
    assert np.abs(scs - want_scs).max() < 1e-4
    assert abs(eer[0] - metrics.calculate_eer(lab, want_scs)[0]) < 0.02        # 200 trials: one swap = 1 %
    assert abs(dcf[0] - metrics.calculate_mindcf(lab, want_scs, p_target=0.05)[0]) < 0.05
    # vectors= short-circuits extraction and reproduces the same numbers exactly
    eer2, dcf2 = sugar_amd.veri_validate(_Loader(trials, 4), model, device=DEV, vectors=vectors)
    assert eer2 == eer and dcf2 == dcf


def test_subnet_sweep_equals_one_at_a_time(meta, supernet):
    """BASELINE config 5 on one rank: evaluate_subnets (next subnet cloned and packed on a host thread while the current
    one runs) gives exactly the numbers of clone -> [batch_forward] -> veri_validate per config, and leaves the
    supernet's active config alone."""
    n_utts = 12
    lens = synth.ragged_lengths(n_utts, lo=24000, hi=40000, seed=13).tolist()
    wavs = synth.ragged_waveforms(lens, seed=13)
    lab, a, b = synth.trial_pairs(n_utts, 120, seed=13)
    trials = _Trials({"label": int(lab[i]), "enroll": "utt%04d.wav" % a[i], "test": "utt%04d.wav" % b[i]}
                     for i in range(len(lab)))
    trials.utterances = _Utts(wavs, [1] * n_utts)
    loader = _Loader(trials, 4)
    recal = [(synth.waveforms(6, 32000, seed=40 + i),) for i in range(2)]
    configs = [cfg_of(meta, n) for n in ("smallest", "rand0", "odd", "rand1")]
    supernet.config = None
    before = supernet.config
    for recal_loader in (None, recal):
        seen = []
        got = sugar_amd.evaluate_subnets(supernet, configs, loader, recal_loader=recal_loader, device=DEV,
                                         on_result=lambda c, r: seen.append(c))
        assert supernet.config == before
        assert len(got) == len(configs) and sorted(seen) == sorted(got)
        for cfg in configs:
            model = _model(supernet, cfg, "bf16")
            supernet.config = None
            if recal_loader is not None:
                sugar_amd.batch_forward(recal_loader, model, device=DEV)
            eer, dcf = sugar_amd.veri_validate(loader, model, device=DEV)
            key = (cfg[0], tuple(cfg[1]), tuple(cfg[2]), cfg[3])
            assert got[key]["eer"] == eer[0] and got[key]["mindcf"] == dcf[0], (recal_loader is not None, cfg, got[key], eer, dcf)


def test_packed_weights_arena_creates_while_busy_and_returns_its_memory(meta, supernet):
    """sugar_subnet_create puts every packed tensor into a few chunks of its own arena and sends them with one copy on
    a private stream (csrc/arena.hip): (a) a subnet packed on a host thread WHILE another model's extraction is queued on
    the GPU gives the vectors of one packed on an idle GPU, bit for bit; (b) creating and releasing 24 subnets leaves
    the device's free memory where it was (the chunks are really freed; the per-tensor hipFree calls are gone); (c) a
    tensor larger than a chunk (the f32 catconv panel of the largest subnet, 9.4 MB) gets a chunk of its own."""
    from concurrent.futures import ThreadPoolExecutor
    from sugar_amd.sweep import _prepare
    dev = torch.device(DEV)
    lens = synth.ragged_lengths(24, lo=24000, hi=40000, seed=21).tolist()
    packed = sugar_amd.PackedUtterances(_Utts(synth.ragged_waveforms(lens, seed=21), [1] * 24), DEV, batch_size=8)
    cfg_a, cfg_b = cfg_of(meta, "rand0"), cfg_of(meta, "largest")

    def vectors_of(sub):
        model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).eval()
        return sugar_amd.extract_vectors(model, packed).matrix.clone()

    idle = _prepare(supernet, cfg_b, "bf16", dev)
    want = vectors_of(idle)
    torch.cuda.synchronize()
    busy_model = _prepare(supernet, cfg_a, "bf16", dev)
    with ThreadPoolExecutor(max_workers=1) as pool:
        for _ in range(3):
            vectors_of(busy_model)                            # enqueued, not waited for
        fut = pool.submit(_prepare, supernet, cfg_b, "bf16", dev)
        for _ in range(3):
            vectors_of(busy_model)
        made_busy = fut.result()
    assert torch.equal(vectors_of(made_busy), want)
    torch.cuda.synchronize()
    for sub in (idle, busy_model, made_busy):
        sub.release()

    f32 = _prepare(supernet, cfg_b, "fp32", dev)              # (c): parity of this path is test_*_fp32's; here it must load
    assert torch.isfinite(vectors_of(f32)).all()
    f32.release()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    for i in range(24):
        sub = _prepare(supernet, cfg_a if i % 2 else cfg_b, "bf16", dev)
        sub.release()
    torch.cuda.synchronize()
    assert abs(torch.cuda.mem_get_info(dev)[0] - free0) < (64 << 20)


def test_concurrent_streams_are_bit_identical(meta, supernet):
    """extract_vectors overlaps consecutive batches on HIP streams, which puts VALU-only kernels (the FFT front-end) and
    MFMA kernels on the same CU.  The vectors must be bit-identical to the one-stream result every time.  (With
    packed-FP32 instructions in the library about 10 % of such runs had one utterance wrong: DESIGN.md section 5b,
    tools/fbank_corun.hip; sugar_amd/build.py switches the feature off.)"""
    import sugar_amd.vectors as V
    n_utts = 48
    lens = synth.ragged_lengths(n_utts, lo=24000, hi=40000, seed=13).tolist()
    packed = sugar_amd.PackedUtterances(_Utts(synth.ragged_waveforms(lens, seed=13), [1] * n_utts), DEV, batch_size=4)
    model = _model(supernet, cfg_of(meta, "odd"), "bf16")
    saved = V.N_LANES
    try:
        V.N_LANES = 1
        ref = sugar_amd.extract_vectors(model, packed).matrix.clone()
        V.N_LANES = 4
        bad = sum(int(not torch.equal(sugar_amd.extract_vectors(model, packed).matrix, ref)) for _ in range(60))
    finally:
        V.N_LANES = saved
    assert bad == 0, "%d of 60 four-stream runs differ from the one-stream vectors" % bad


def test_full_size_batch256_properties(supernet):
    """BASELINE config 2 size (256 x 3 s, Base, bf16): size-independent properties instead of a CPU oracle run."""
    model = _model(supernet, layout.BASE, "bf16")
    wav = synth.waveforms(8, 48000, seed=21)
    big = wav.repeat(32, 1).to(DEV)                       # 256 utterances = 32 copies of 8 distinct ones
    emb = model(big)
    assert emb.shape == (256, 192) and torch.isfinite(emb).all()
    # (1) batch invariance: every copy of an utterance gets the same embedding as in a batch of 8
    small = model(wav.to(DEV))
    assert (emb.view(32, 8, 192) - small.unsqueeze(0)).abs().max() < 1e-2
    # (2) the 8 distinct utterances match the oracle within the bf16 tolerance
    osd = ref_cpu.clone_state(supernet.state_dict(), layout.BASE)
    want = ref_cpu.embed(osd, 3, wav)
    assert _cos(small, want).min() >= 0.999
    # ... and so does every one of the 256 rows of the big batch, directly (R = 77 056 frames: the kernels and grids the
    # bench runs -- full rounds of 256 x 256 ping-pong tiles in catconv, the weights-in-registers GEMM in the blocks)
    want256 = want.repeat(32, 1)
    assert _cos(emb, want256).min() >= 0.999 and _relerr(emb, want256) < REL["bf16"]
    # (3) determinism: same input, same bits
    assert torch.equal(model(big), emb)
    # (4) normalised output has unit length and the same direction
    en = model(big, normalize=True)
    assert (en.norm(dim=1) - 1).abs().max() < 1e-5
    assert F.cosine_similarity(en, emb, dim=1).min() > 0.99999


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_sampled_subnets_match_oracle(precision, supernet, sup_state):
    """24 subnets drawn by the reference's sampler (TDNNPath over DynamicTDNN.space(), dynamictdnn.py:792-879: depth
    2-4, channels 128-512 in the space's steps, kernels 1/3/5, catconv 384-1536) on a small ragged batch, against the
    CPU oracle: shapes no golden vector covers (the goldens hold 7 named subnets).  5 utterances < #CUs, so the Res2
    chain runs under its few-utterance segment rule as well."""
    sampler = sugar_amd.TDNNPath(supernet.space(verbose=False))
    lens = [16000, 23456, 9999, 48000, 31000]
    wavs = synth.ragged_waveforms(lens, seed=77)
    worst_c, worst_r = 1.0, 0.0
    for k in range(24):
        cfg = sampler.sample()
        model = _model(supernet, cfg, precision)
        emb = model([w.to(DEV) for w in wavs])
        want = ref_cpu.embed_ragged(ref_cpu.clone_state(sup_state, cfg), cfg[0], wavs)
        c, r = _cos(emb, want).min().item(), _relerr(emb, want)
        worst_c, worst_r = min(worst_c, c), max(worst_r, r)
        assert c >= TOL[precision] and r < REL[precision], (k, cfg, c, r)
    print("sampled subnets", precision, "worst cosine %.6f  worst rel L2 %.2e" % (worst_c, worst_r))


def test_host_fed_list_equals_resident_list(supernet):
    """extract_vectors over a HOST-resident list (pinned triple-buffered staging on a copy stream, packed by n_jobs
    threads) gives bit-identical vectors to the same list on the GPU and to PackedUtterances, over more batches than
    staging slots (slot reuse), for 1 and 3 packing threads, float32 and float64 inputs."""
    model = _model(supernet, layout.SMALLEST, "bf16")
    n_utts = 37
    lens = synth.ragged_lengths(n_utts, lo=20000, hi=70000, seed=21).tolist()
    wavs = synth.ragged_waveforms(lens, seed=21)
    segs = [1 + (i % 3 == 1) for i in range(n_utts)]
    host = _Utts(wavs, segs)
    resident = _Utts([w.to(DEV) for w in wavs], segs)
    want = sugar_amd.extract_vectors(model, resident, norm=True, device=DEV, batch_size=4)
    for jobs in (1, 3):
        got = sugar_amd.extract_vectors(model, host, norm=True, device=DEV, batch_size=4, n_jobs=jobs)
        assert list(got.keys()) == list(want.keys())
        assert torch.equal(got.matrix, want.matrix), jobs
    packed = sugar_amd.extract_vectors(model, sugar_amd.PackedUtterances(host, DEV, batch_size=4), norm=True)
    assert torch.equal(packed.matrix, want.matrix)
    host64 = _Utts([w.double() for w in wavs], segs)
    assert torch.equal(sugar_amd.extract_vectors(model, host64, norm=True, device=DEV, batch_size=4).matrix, want.matrix)
