"""
Deterministic synthetic data: supernet weights, BN statistics, waveforms, trial lists.

There is no network (no HuggingFace checkpoint, no VoxCeleb), so every test and
benchmark runs on synthetic weights in the reference's own state-dict layout
(keys as produced by ``tdnn8m2g`` in /root/reference/sugar/models/dynamictdnn.py:894-936)
and on N(0,1) waveforms (the README uses ``torch.randn(1, 10000)``, README.md:67).

The fill is a counter hash (splitmix64 of ``fnv1a(name) + index``), not torch's
RNG, so this container and the GPU box regenerate identical tensors without
shipping 30 MB files.  BN running statistics, ``k53``/``k31`` kernel-transform
matrices and biases are deliberately *non-trivial* (defaults mu=0, var=1, k53=I
would hide BN / kernel-transform bugs).
"""
import math

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for c in name.encode():
        h ^= c
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 values in [0, 1), a pure function of (name, seed, index)."""
    base = np.uint64((_fnv1a64(name) + 0x632BE59BD9B4E019 * (seed + 1)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + base
    z = _splitmix64(idx)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def uniform(name, shape, lo, hi, seed=0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, n, seed)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def normal(name, shape, seed=0) -> torch.Tensor:
    """N(0,1) float32 by Box-Muller on two counter-hash streams."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(name + "/u1", n, seed)
    u2 = uniform01(name + "/u2", n, seed)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return torch.from_numpy((r * np.cos(2.0 * math.pi * u2)).astype(np.float32).reshape(shape))


def waveforms(batch: int, num_samples: int, seed: int = 0, first: int = 0) -> torch.Tensor:
    """[batch, num_samples] N(0,1) float32; utterance ``first+i`` is independent of the batch it is drawn in."""
    return torch.stack([normal("wav/%d" % (first + i), (num_samples,), seed) for i in range(batch)])


def ragged_waveforms(lengths, seed: int = 0):
    """List of 1-D N(0,1) waveforms with the given sample counts (BASELINE config 4: 2-20 s)."""
    return [normal("wav/%d" % i, (int(n),), seed) for i, n in enumerate(lengths)]


def ragged_lengths(n_utts: int, lo: int = 32000, hi: int = 320000, seed: int = 0) -> np.ndarray:
    u = uniform01("lengths", n_utts, seed)
    return (lo + np.floor(u * (hi - lo + 1))).astype(np.int64)


# The first fill of round 1 (conv gain sqrt(2), pooling.linear1 x 1, pooling.linear2 x 4): activations grow x130 through
# the depth-4 supernet, tanh saturates and the softmax over time is nearly one-hot -- the regime in which the bf16 path
# falls below the 0.999 cosine bar (DESIGN.md "bf16 sensitivity").  Kept as a fixture: tests/golden/round2.npz holds the
# reference's embeddings for it, tests/test_bf16_sensitivity.py sweeps from the shipped fill (t = 0) to this one (t = 1).
ILL_CONDITIONED = {"conv_gain": math.sqrt(2.0), "pool1_gain": 1.0, "pool2_gain": 4.0}


def conditioning(t: float) -> dict:
    """Fill gains on the geometric path from the shipped, trained-like fill (t = 0) to ILL_CONDITIONED (t = 1)."""
    return {"conv_gain": 2.0 ** (0.5 * t), "pool1_gain": 0.35 ** (1.0 - t), "pool2_gain": 2.8 ** (1.0 - t) * 4.0 ** t}


def fill_state_dict(shapes: dict, seed: int = 0, conv_gain: float = 1.0, pool1_gain: float = 0.35,
                    pool2_gain: float = 2.8) -> dict:
    """
    Fill a {key: shape} layout (supernet or cloned-subnet naming) with deterministic values.

    Rules by key (names follow the reference modules, dynamicarch.py:363-399, dynamictdnn.py:628-638):
      *k53 / *k31                : identity + U(-0.15, 0.15)   (reference initialises to identity, dynamicarch.py:385-392)
      *running_mean              : U(-0.3, 0.3)
      *running_var               : U(0.5, 1.5)
      *num_batches_tracked       : 0 (int64)
      BN *.weight (1-d)          : U(0.5, 1.5);  BN/linear *.bias : U(-0.2, 0.2)
      conv *.weight  [O, I, k]   : U(+-sqrt(3 / (I k)))       (unit gain: activations stay O(1) through the depth,
                                   as BN recalibration keeps them in a trained supernet); pooling.linear1 x 0.35 and
                                   pooling.linear2 x 2.8 put the attention in a trained-like regime: tanh unsaturated,
                                   softmax over time clearly non-uniform (effective frames ~100-170 of 301) but not
                                   one-hot.  (With saturated tanh and one-hot softmax the pooled statistics become a
                                   discontinuous function of the activations and no reduced-precision path can track
                                   the f32 reference; see DESIGN.md "bf16 sensitivity".)  conv_gain / pool1_gain /
                                   pool2_gain move these three scales (defaults: the shipped fill).
      linear *.weight [O, I]     : U(+-sqrt(3 / I))
    """
    out = {}
    for key, shape in shapes.items():
        shape = tuple(shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf in ("k53", "k31", "k75"):
            n = shape[0]
            out[key] = torch.eye(n) + uniform(key, shape, -0.15, 0.15, seed)
        elif leaf == "running_mean":
            out[key] = uniform(key, shape, -0.3, 0.3, seed)
        elif leaf == "running_var":
            out[key] = uniform(key, shape, 0.5, 1.5, seed)
        elif leaf == "num_batches_tracked":
            out[key] = torch.zeros(shape, dtype=torch.int64)
        elif leaf == "bias":
            out[key] = uniform(key, shape, -0.2, 0.2, seed)
        elif leaf == "weight" and len(shape) == 1:
            out[key] = uniform(key, shape, 0.5, 1.5, seed)
        elif leaf == "weight" and len(shape) == 3:
            a = math.sqrt(3.0 / (shape[1] * shape[2]))
            if "pooling.linear1" in key:
                a *= pool1_gain
            elif "pooling.linear2" in key:
                a *= pool2_gain
            else:
                a *= conv_gain
            out[key] = uniform(key, shape, -a, a, seed)
        elif leaf == "weight" and len(shape) == 2:
            a = math.sqrt(3.0 / shape[1])
            out[key] = uniform(key, shape, -a, a, seed)
        elif leaf == "W":  # loss head (module.__L__.W) when present
            out[key] = uniform(key, shape, -0.05, 0.05, seed)
        else:
            raise KeyError("no fill rule for %s %s" % (key, shape))
    return out


def trial_pairs(n_utts: int, n_trials: int, seed: int = 0):
    """Synthetic verification trials: (labels, enroll_idx, test_idx); enroll != test."""
    a = np.floor(uniform01("trial/enroll", n_trials, seed) * n_utts).astype(np.int64)
    b = np.floor(uniform01("trial/test", n_trials, seed) * (n_utts - 1)).astype(np.int64)
    b = np.where(b >= a, b + 1, b)
    lab = (uniform01("trial/label", n_trials, seed) < 0.5).astype(np.int64)
    return lab, a, b
