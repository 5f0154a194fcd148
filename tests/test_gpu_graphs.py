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
