"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Seeded inputs of the norm2 / list-of-modes scoring fixture (shared by the generator,
which needs /root/reference, and by the tests, which do not): 40 utterances of 1-3 segment vectors, 300 trials.
"""
import torch

from sugar_amd import synth

N_UTTS, N_TRIALS, DIM = 40, 300, 192


def inputs():
    counts = [1 + (i * 5) % 3 for i in range(N_UTTS)]
    vectors = []
    for i, c in enumerate(counts):
        v = synth.normal("norm2/utt%d" % i, (c, DIM), seed=31)
        vectors.append(v if i % 4 else torch.nn.functional.normalize(v, dim=1))     # unit and non-unit vectors
    lab, a, b = synth.trial_pairs(N_UTTS, N_TRIALS, seed=31)
    return vectors, lab, a, b
