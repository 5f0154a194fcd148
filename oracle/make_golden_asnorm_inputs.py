"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Seeded inputs shared by oracle/make_golden_asnorm.py and the tests."""
import numpy as np

from sugar_amd import synth

N_UTTS, N_COHORT, N_TRIALS, DIM = 40, 700, 500, 192
NTOPS = (300, 120)     # the reference raises for nTop < 0 (torch.topk with a negative k, scores.py:372)


def inputs():
    """({utt: Tensor [n_seg, 192]}, cohorts Tensor [700, 192], trial scores float64 [500], enroll names, test names)."""
    vectors = {}
    for i in range(N_UTTS):
        base = synth.normal("asnorm/spk%d" % (i % 8), (1, DIM), seed=5)          # 8 "speakers": correlated vectors
        vectors["utt%03d.wav" % i] = base + 0.6 * synth.normal("asnorm/utt%d" % i, (1 + i % 3, DIM), seed=5)
    cohorts = synth.normal("asnorm/cohort", (N_COHORT, DIM), seed=6)
    cohorts[:64] += synth.normal("asnorm/spk0", (1, DIM), seed=5)                 # part of the cohort near speaker 0
    cohorts[5] = cohorts[4]                                                       # exact ties in every row
    rs = np.random.RandomState(11)
    e = rs.randint(0, N_UTTS, N_TRIALS)
    t = rs.randint(0, N_UTTS, N_TRIALS)
    scores = rs.randn(N_TRIALS) * 0.2 + 0.1
    return vectors, cohorts, scores, ["utt%03d.wav" % i for i in e], ["utt%03d.wav" % i for i in t]
