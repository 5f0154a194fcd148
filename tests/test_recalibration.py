"""
BN recalibration (batch_forward, sugar/models/__init__.py:201-242).

tests/golden/recal.npz holds what the REAL reference subnet produced (oracle/make_golden_recal.py): the running
statistics of every BatchNorm after train()-mode forwards over seeded batches, the train-mode embeddings, and an
eval() forward afterwards.  CPU: the oracle reproduces them.  GPU: sugar_amd.batch_forward through the C-ABI does.
"""
import os

import numpy as np
import pytest
import torch

import sugar_amd
from oracle import ref_cpu
from oracle.make_golden_recal_inputs import CASES, EVAL_SEED, batches
from sugar_amd import layout, synth

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "recal.npz"))


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_train_mode_matches_reference(name):
    cfg, B, L, nb = CASES[name]
    osd = ref_cpu.clone_state(synth.fill_state_dict(layout.supernet_shapes(), seed=0), cfg)
    for bi, wav in enumerate(batches(name, nb, B, L)):
        e = ref_cpu.subnet_forward(osd, cfg[0], ref_cpu.logmel_fbank(wav).squeeze(1), train_momentum=0.1)
        assert _rel(e.numpy(), GOLD["%s/train_emb%d" % (name, bi)]) < 1e-4
    n = 0
    for k in GOLD.files:
        if k.startswith(name + "/") and (k.endswith("running_mean") or k.endswith("running_var")):
            assert _rel(osd[k[len(name) + 1:]].numpy(), GOLD[k]) < 1e-5, k
            n += 1
    assert n == 2 * (3 + 9 * cfg[0])
    e = ref_cpu.embed(osd, cfg[0], synth.waveforms(3, L, seed=EVAL_SEED))
    assert _rel(e.numpy(), GOLD[name + "/eval_emb"]) < 1e-4


class _Loader(list):
    pass


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol_stat,tol_emb", [("fp32", 2e-4, 5e-4), ("bf16", 3e-2, 6e-2)])
@pytest.mark.parametrize("name", sorted(CASES))
def test_batch_forward_matches_reference(name, precision, tol_stat, tol_emb):
    cfg, B, L, nb = CASES[name]
    dev = torch.device("cuda", 0)
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    sub = sup.clone(cfg).set_precision(precision)
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).eval()
    before = model(synth.waveforms(3, L, seed=EVAL_SEED).to(dev)).cpu().numpy()

    # train()-mode forwards one by one: embeddings of each recalibration step
    model.train()
    for bi, wav in enumerate(batches(name, nb, B, L)):
        e = model(wav.to(dev))
        assert _rel(e.cpu().numpy(), GOLD["%s/train_emb%d" % (name, bi)]) < tol_emb
    model.eval()
    sd = sub.state_dict()                                  # syncs the statistics into the module's buffers
    worst = 0.0
    for k in GOLD.files:
        if k.startswith(name + "/") and (k.endswith("running_mean") or k.endswith("running_var")):
            worst = max(worst, _rel(sd[k[len(name) + 1:]].numpy(), GOLD[k]))
    assert worst < tol_stat, worst
    assert int(sd["layer1.bn.num_batches_tracked"]) == nb
    after = model(synth.waveforms(3, L, seed=EVAL_SEED).to(dev)).cpu().numpy()
    assert _rel(after, GOLD[name + "/eval_emb"]) < tol_emb
    assert _rel(after, before) > 1e-3                      # the eval path really uses the new statistics

    # the batch_forward call itself (fresh clone): same statistics, model mode restored
    sub2 = sup.clone(cfg).set_precision(precision)
    model2 = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub2, transform=sugar_amd.LogMelFbanks())).eval()
    loader = _Loader((w, None) for w in batches(name, nb, B, L))
    sugar_amd.batch_forward(loader, model2, device="cuda:0")
    assert not model2.training
    sd2 = sub2.state_dict()
    for k in sd:
        assert torch.equal(sd[k], sd2[k]), k               # deterministic: bit-identical statistics
    bn = sugar_amd.bn_state_dict(sub2)
    assert "layer1.bn.running_var" in bn


@pytest.mark.gpu
def test_reset_bn_and_errors():
    dev = torch.device("cuda", 0)
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    sub = sup.clone(layout.SMALLEST).set_precision("fp32")
    sub.reset_bn()
    sd = sub.state_dict()
    assert float(sd["bnembed.running_mean"].abs().max()) == 0.0 and float(sd["bnpool.running_var"].min()) == 1.0
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev)
    model.train()
    with pytest.raises(ValueError):                        # torch: "Expected more than 1 value per channel"
        model(synth.waveforms(1, 8000, seed=1).to(dev))
    smodel = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sup, transform=sugar_amd.LogMelFbanks())).to(dev).train()
    with pytest.raises(sugar_amd._lib.SugarError):         # supernet in train(): clone first
        smodel(synth.waveforms(2, 8000, seed=1).to(dev))


@pytest.mark.gpu
@pytest.mark.parametrize("precision,wreg_min_r", [("fp32", None), ("bf16", None), ("bf16", 0)])
def test_conv_epilogue_statistics_equal_the_separate_pass(precision, wreg_min_r):
    """The generic conv kernel's epilogue leaves the BatchNorm partial sums behind (ConvIO::stat_part); with
    recal_separate_stats=1 every BatchNorm runs bn_stats_kernel over the stored tensor instead.  Both sum the SAME stored
    values (f32 partials of 128 rows, then doubles, against doubles throughout), on a ragged batch whose last row tile is
    partial.  f32: every running statistic and the embeddings agree to f32 rounding.  bf16: the first BatchNorm fed by a
    fused conv (Res2 step 0 of block 0; layer1 and first_conv run on the LDS-DMA GEMMs in both modes) agrees to f32
    rounding; behind it a 1e-7 change of a scale moves bf16 roundings of the normalised tensor, so later layers are held
    to the bf16 path's own noise level only.  gemm_wreg_min_r=0 sends layer1 and the 512-channel convs to the
    weights-in-registers GEMM whatever the batch size, whose epilogue keeps its own running sums (the default threshold
    is 19 200 frames; these 1 285 frames would stay on the tile kernels and their separate pass)."""
    dev = torch.device("cuda", 0)
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    lens = [16000, 23456, 48000, 9999, 31000]
    wavs = [synth.waveforms(1, n, seed=40 + i)[0].to(dev) for i, n in enumerate(lens)]
    out = {}
    for sep in (0, 1):
        sub = sup.clone(layout.BASE).set_precision(precision)
        spk = sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks()).to(dev)
        spk.train()
        knobs = {"recal_separate_stats": sep}
        if wreg_min_r is not None:
            knobs["gemm_wreg_min_r"] = wreg_min_r
        with sugar_amd._lib.tuned(**knobs):
            emb = spk.embed_ragged(wavs, normalize=False)
        spk.eval()
        out[sep] = (emb.cpu().numpy(), {k: v.clone() for k, v in sub.state_dict().items() if "running" in k})
    tight, loose = 2e-6, (2e-6 if precision == "fp32" else 2e-2)
    first = ("layer1.bn.", "blocks.0.first_conv.bn.", "blocks.0.resconv.convs.0.bn.")
    assert _rel(out[0][0], out[1][0]) < 50 * loose
    for k in out[0][1]:
        assert _rel(out[0][1][k].numpy(), out[1][1][k].numpy()) < (tight if k.startswith(first) else loose), k


@pytest.mark.gpu
def test_batch_forward_host_batches_equal_resident_batches():
    """batch_forward stages DataLoader batches from pageable host memory through three pinned slots and a copy stream
    (models._BatchFeed); seven batches reuse every slot at least twice.  Running statistics are bit-equal to the same
    batches already on the GPU, for float32 batches, for a batch the feed must convert (float64) and a [B, 1, L] one."""
    dev = torch.device("cuda", 0)
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    host = [synth.waveforms(5, 20000 + 1600 * i, seed=60 + i) for i in range(7)]
    host[2] = host[2].double()
    host[4] = host[4].unsqueeze(1)
    stats = {}
    for name, loader in (("resident", [(b.to(dev),) for b in host]), ("host", [(b,) for b in host])):
        sub = sup.clone(layout.SMALLEST).set_precision("bf16")
        model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).eval()
        sugar_amd.batch_forward(loader, model, device="cuda:0")
        stats[name] = {k: v.clone() for k, v in sub.state_dict().items() if "running" in k or "tracked" in k}
    assert int(stats["host"]["layer1.bn.num_batches_tracked"]) == 7
    for k in stats["host"]:
        assert torch.equal(stats["host"][k], stats["resident"][k]), k


@pytest.mark.gpu
def test_lazy_bn_read_back_changes_nothing_but_when_the_buffers_are_read():
    """TDNNSubnet.bn_lazy (the NAS sweep sets it on the subnets it evaluates and throws away): after recalibration the
    statistics are folded into the eval path at once, the module's buffers are brought up to date only when someone asks
    (state_dict, another precision).  Eval embeddings and the final state dict are bit-equal to the eager mode."""
    dev = torch.device("cuda", 0)
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    loader = [(synth.waveforms(5, 24000, seed=70 + i),) for i in range(3)]
    wav = synth.waveforms(3, 20000, seed=9).to(dev)
    got = {}
    for lazy in (False, True):
        sub = sup.clone(layout.SMALLEST).set_precision("bf16")
        sub.bn_lazy = lazy
        model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).eval()
        before = float(sub.layer1.bn.running_mean.abs().sum())
        sugar_amd.batch_forward(loader, model, device="cuda:0")
        emb = model(wav).clone()
        stale = float(sub.layer1.bn.running_mean.abs().sum()) == before      # buffers untouched so far?
        assert stale == lazy
        sd = {k: v.clone() for k, v in sub.state_dict().items()}            # asks for the buffers
        assert float(sub.layer1.bn.running_mean.abs().sum()) != before
        emb32 = model.module.__S__.set_precision("fp32") and model(wav).clone()   # another handle, packed from the buffers
        got[lazy] = (emb, sd, emb32)
    assert torch.equal(got[True][0], got[False][0]) and torch.equal(got[True][2], got[False][2])
    for k in got[False][1]:
        assert torch.equal(got[True][1][k], got[False][1][k]), k
