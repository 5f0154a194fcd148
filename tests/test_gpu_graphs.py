"""
GPU: the embedding forward captured into a HIP graph (sugar_amd/graphs.py) replays bit-identically to the eager call,
for fresh inputs, in both precisions, and refuses shapes it was not captured for.
"""
import pytest
import torch

import sugar_amd
from sugar_amd import _lib, layout, synth
from sugar_amd.graphs import GraphedEmbedder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(supernet, cfg, precision):
    sub = supernet.clone(cfg).set_precision(precision)
    supernet.config = None
    return sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("B,L", [(1, 48000), (4, 32000)])
def test_graph_replay_is_bit_identical_to_eager(B, L, precision, supernet):
    model = _model(supernet, layout.BASE, precision)
    g = GraphedEmbedder(model, synth.waveforms(B, L, seed=1).to(DEV), normalize=True)
    for seed in (2, 3, 4):
        wav = synth.waveforms(B, L, seed=seed).to(DEV)
        got = g(wav).clone()
        ref = model(wav, normalize=True)
        torch.cuda.synchronize()
        assert got.shape == (B, 192)
        assert torch.equal(got, ref), (got - ref).abs().max().item()
    with pytest.raises(_lib.SugarError):
        g(synth.waveforms(B + 1, L, seed=5).to(DEV))


def test_graph_is_never_replayed_on_released_weights(supernet):
    """A captured graph bakes in the packed-weight handle's device pointers and the cached workspace.  load_state_dict
    (its post-hook releases the handle), release() and set_precision all invalidate them: the next call must capture
    again (default) or raise -- never replay kernels that read freed memory."""
    model = _model(supernet, layout.SMALLEST, "bf16")
    sub = model.module._subnet()
    wav = synth.waveforms(2, 32000, seed=7).to(DEV)
    g = GraphedEmbedder(model, wav, normalize=True)
    first = g(wav).clone()
    assert g.captures == 1
    # new weights through load_state_dict: the replay must show THEM, not the stale ones
    sd = {k: v.clone() for k, v in sub.state_dict().items()}
    sd["linear.bias"] = sd["linear.bias"] + 0.5
    sub.load_state_dict(sd)
    got = g(wav).clone()
    ref = model(wav, normalize=True)
    torch.cuda.synchronize()
    assert g.captures == 2
    assert torch.equal(got, ref) and not torch.equal(got, first)
    # release() (what a sweep does with a finished subnet) and a precision change
    sub.release()
    assert torch.equal(g(wav), model(wav, normalize=True)) and g.captures == 3
    sub.set_precision("fp32")
    got32 = g(wav).clone()
    assert g.captures == 4 and torch.equal(got32, model(wav, normalize=True))
    assert g(wav) is g.static_out and g.captures == 4          # nothing changed: plain replay
    # on_stale="raise"
    g2 = GraphedEmbedder(model, wav, normalize=True, on_stale="raise")
    g2(wav)
    sub.release()
    with pytest.raises(_lib.SugarError):
        g2(wav)
