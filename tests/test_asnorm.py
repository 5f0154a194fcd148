"""
Score-normalisation back-end (score_cohorts / asnorm / snorm, sugar/scores.py:77-111, 269-400).

tests/golden/asnorm.npz holds the outputs of the reference's own functions on seeded inputs
(oracle/make_golden_asnorm.py).  CPU: the oracle reproduces them.  GPU: sugar_amd's kernels do, through the C-ABI.
"""
import os

import numpy as np
import pytest
import torch

import sugar_amd
from oracle import ref_cpu
from oracle.make_golden_asnorm_inputs import N_COHORT, NTOPS, inputs

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "asnorm.npz"))


def test_oracle_matches_reference():
    vectors, cohorts, scores, enr, tst = inputs()
    cs = ref_cpu.score_cohorts(cohorts, vectors)
    assert np.array_equal(np.stack([cs[k] for k in vectors]), GOLD["cohort_scores"])
    for nt in NTOPS:
        assert np.allclose(ref_cpu.asnorm(scores, enr, tst, cs, nt), GOLD["asnorm_%d" % nt], rtol=1e-6, atol=1e-6)
    assert np.allclose(ref_cpu.snorm(scores, enr, tst, cs), GOLD["snorm"], rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_score_cohorts_and_norms_match_reference():
    vectors, cohorts, scores, enr, tst = inputs()
    cs = sugar_amd.score_cohorts(cohorts, {k: v.cuda() for k, v in vectors.items()}, device="cuda:0")
    assert list(cs.keys()) == list(vectors.keys())
    got = np.stack([cs[k] for k in vectors])
    assert got.shape == (len(vectors), N_COHORT) and got.dtype == np.float32
    assert np.abs(got - GOLD["cohort_scores"]).max() < 2e-6
    trials = {"score": scores, "enroll": enr, "test": tst}
    for nt in NTOPS:
        out = sugar_amd.asnorm(trials, cs, nTop=nt)
        assert out.dtype == np.float64
        assert np.abs(out - GOLD["asnorm_%d" % nt]).max() < 2e-4, nt      # |normalised scores| reach ~30
        assert np.array_equal(np.argsort(out, kind="stable"), np.argsort(GOLD["asnorm_%d" % nt], kind="stable")) or \
            np.abs(np.sort(out) - np.sort(GOLD["asnorm_%d" % nt])).max() < 2e-4
    assert np.abs(sugar_amd.snorm(trials, cs) - GOLD["snorm"]).max() < 2e-4
    # a plain {utterance: ndarray} cohort dict (the reference's return type) is accepted as well
    plain = {k: np.array(cs[k]) for k in vectors}
    assert np.array_equal(sugar_amd.asnorm(trials, plain, nTop=300), sugar_amd.asnorm(trials, cs, nTop=300))
    # nTop < 0 ("most dissimilar", documented at scores.py:348 but raising there): against the oracle restatement
    ocs = ref_cpu.score_cohorts(cohorts, vectors)
    assert np.abs(sugar_amd.asnorm(trials, cs, nTop=-50) - ref_cpu.asnorm(scores, enr, tst, ocs, -50)).max() < 2e-4


@pytest.mark.gpu
def test_topk_stats_exact_selection_with_ties():
    """The radix select must return torch.topk's multiset: rows full of ties, negatives, k = 1 and k = n."""
    rs = np.random.RandomState(3)
    rows = [np.round(rs.randn(1000) * 4) / 4,                 # ~30 distinct values: massive ties
            -np.abs(rs.randn(1000)),                          # all negative
            np.r_[np.zeros(990), rs.randn(10)],               # zeros at the threshold
            rs.randn(1000)]
    mat = torch.tensor(np.stack(rows), dtype=torch.float32, device="cuda:0")
    lib = sugar_amd._lib.lib()
    for k in (1, 7, 300, 1000, -1, -300):
        mean = torch.empty(4, device="cuda:0")
        std = torch.empty(4, device="cuda:0")
        sugar_amd._lib.check(lib.sugar_topk_stats(mat.data_ptr(), 4, 1000, k, float(abs(k)), mean.data_ptr(),
                                                  std.data_ptr(), torch.cuda.current_stream().cuda_stream))
        for r, row in enumerate(rows):
            v = np.sort(row.astype(np.float32))
            sel = v[:abs(k)] if k < 0 else v[len(v) - k:]
            m = sel.astype(np.float64).mean()
            s = np.sqrt(((sel.astype(np.float64) - m) ** 2).sum() / abs(k))
            assert abs(float(mean[r]) - m) < 1e-6, (k, r)
            assert abs(float(std[r]) - s) < 1e-6, (k, r)
    with pytest.raises(ValueError):                            # torch.topk: k out of range
        sugar_amd._lib.check(lib.sugar_topk_stats(mat.data_ptr(), 4, 1000, 1001, 1001.0, mean.data_ptr(),
                                                  std.data_ptr(), torch.cuda.current_stream().cuda_stream))
