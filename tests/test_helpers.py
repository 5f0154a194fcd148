"""
The small helpers on sugar's scoring / vector call surface (sugar/scores.py:27-74, 219-266, 402-420; sugar/vectors.py:25-60,
157-178) against outputs of the reference's own functions (tests/golden/helpers.npz, oracle/make_golden_helpers.py), on
the asnorm fixture's inputs.  Host-side code as in the reference; score_batch runs on the GPU.
"""
import os

import numpy as np
import pandas as pd
import pytest
import torch

import sugar_amd
from oracle.make_golden_asnorm_inputs import inputs

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "helpers.npz"))


def test_ztnorm_matches_reference():
    vectors, cohorts, scores, enr, tst = inputs()
    keys = list(vectors.keys())
    cohort = {k: GOLD["cohort_scores"][i] for i, k in enumerate(keys)}         # the reference's own cohort scores
    df = pd.DataFrame({"score": scores, "enroll": enr, "test": tst})
    for nt in (None, 300, -300):
        got = sugar_amd.ztnorm(df, cohort, nTop=nt)
        assert got.dtype == np.float64
        # bar: 1e-6 relative.  The reference's row-wise pandas apply hands python floats to numpy.float32 statistics, which
        # numpy >= 2 (NEP 50) evaluates in float32; here the same formula runs in float64
        want = GOLD["ztnorm_%s" % nt]
        assert np.abs(got - want).max() < 1e-6 * max(1.0, np.abs(want).max()), nt
    assert np.array_equal(sugar_amd.ztnorm({"score": scores, "enroll": enr, "test": tst}, cohort), sugar_amd.ztnorm(df, cohort))


def test_normalize_scores_and_sctr_files(tmp_path):
    vectors, cohorts, scores, enr, tst = inputs()
    assert np.abs(sugar_amd.normalize_scores(scores, -1.0, 2.0) - GOLD["normalize_scores"]).max() < 1e-12
    pairs = [[e, t] for e, t in zip(enr, tst)]
    path = str(tmp_path / "scores.txt")
    sugar_amd.save_sctr(scores[:20], pairs[:20], path, root_dir="utt", scale=True)
    assert open(path, "rb").read() == GOLD["sctr_text"].tobytes()              # byte for byte what the reference wrote
    sc, tr = sugar_amd.load_sctr(path)
    assert np.array_equal(np.array(sc), GOLD["sctr_scores"]) and tr[0] == [pairs[0][0].replace("utt", ""), pairs[0][1].replace("utt", "")]
    with pytest.raises(AssertionError):
        sugar_amd.save_sctr(scores[:2], pairs[:2], path)                      # refuses to overwrite, as upstream


def test_length_normalize_and_vector_transform():
    vectors, *_ = inputs()
    v = vectors["utt005.wav"]
    for su in (False, True):
        n, avg, std = sugar_amd.length_normalize(v.clone(), scaleup=su)
        assert torch.allclose(n, torch.from_numpy(GOLD["lennorm_%d" % su]), rtol=1e-6, atol=1e-7)
        assert np.allclose([float(avg), float(std)], GOLD["lennorm_%d_stats" % su], rtol=1e-5, atol=1e-6)
    n, avg, std = sugar_amd.length_normalize(v.numpy().copy(), scaleup=True)
    assert isinstance(n, np.ndarray) and np.allclose(n, GOLD["lennorm_np"], rtol=1e-6, atol=1e-7)
    assert np.allclose([float(avg), float(std)], GOLD["lennorm_np_stats"], rtol=1e-5, atol=1e-6)
    sub = {k: vectors[k] for k in list(vectors)[:3]}
    vt = sugar_amd.vector_transform(sub, transform=lambda x: x * 2.0)
    assert list(vt) == list(sub) and np.array_equal(np.concatenate([vt[k].numpy() for k in vt]), GOLD["vector_transform"])
    cp = sugar_amd.vector_transform(sub)
    assert cp is not sub and all(cp[k] is sub[k] for k in sub)


@pytest.mark.gpu
def test_score_batch_matches_reference():
    vectors, cohorts, *_ = inputs()
    got = sugar_amd.score_batch(vectors["utt002.wav"], cohorts[:50], device="cuda:0")
    assert got.shape == GOLD["score_batch"].shape and got.dtype == np.float32
    assert np.abs(got - GOLD["score_batch"]).max() < 2e-6
    with pytest.raises(ValueError):
        sugar_amd.score_batch(vectors["utt002.wav"], cohorts[:50], mode="norm2")
