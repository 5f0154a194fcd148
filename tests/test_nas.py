"""NAS tooling of SURVEY 8f rank 4 (sugar_amd/nas.py) against what the reference's own classes returned
(tests/golden/nas.npz, made by oracle/make_golden_nas.py).  decode / fit / ea_solve are PARITY UNPINNED (reasons in
nas.py's header) and are checked by round trips, by what the predictor learns, and against exhaustive search."""
import os

import numpy as np
import pytest
import torch

from sugar_amd import nas

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "nas.npz"))
CSPACE = [[2, 3, 4]] + [[128, 176, 256, 384, 512]] * 5 + [[1, 3, 5]] * 5 + [[384, 536, 768, 1152, 1536]]
DIMS = ["depth"] + ["width%d" % i for i in range(5)] + ["kernel%d" % i for i in range(5)] + ["cat"]
DCONFIG = {d: CSPACE[i] for i, d in enumerate(DIMS)}
CONFIGS = [[int(v) for v in row if v >= 0] for row in G["configs"]]


def _codec(mode=None):
    return nas.ArchCoder(len(CSPACE), DCONFIG, nas.config2d, mode=mode)


@pytest.mark.parametrize("mode", [None, "b", "c"])
def test_archcoder_encode_matches_reference(mode):
    codec = _codec(mode)
    name = mode or "onehot"
    assert codec.nfeat == int(G["nfeat/" + name])
    got = np.stack([codec.encode(c) for c in CONFIGS])
    assert got.dtype == np.float32 and np.array_equal(got, G["encode/" + name])


@pytest.mark.parametrize("mode", [None, "b", "c"])
def test_archcoder_decode_round_trips(mode):
    """The reference's decode raises on numpy >= 1.24 (np.int; recorded as decode_raises in the fixture): parity unpinned."""
    assert int(G["decode_raises"]) == 1
    codec = _codec(mode)
    for c in CONFIGS:
        assert codec.decode(codec.encode(c), 1) == c
        d = codec.decode(codec.encode(c), 0)
        assert d == nas.config2d(c)


def test_config2d_matches_reference():
    d = nas.config2d(CONFIGS[0])
    assert list(d.keys()) == [str(k) for k in G["config2d_keys"]]
    assert [-1 if v is None else v for v in d.values()] == G["config2d_vals"].tolist()
    with pytest.raises(AssertionError):
        nas.config2d([3, 128, 128, 1, 1, 384])


def _golden_predictor():
    p = nas.AccuracyPredictor(int(G["nfeat/onehot"]))
    p.load_state({k[len("pred/model/"):]: G[k] for k in G.files if k.startswith("pred/model/")},
                 {k[len("pred/ynorm/"):]: G[k] for k in G.files if k.startswith("pred/ynorm/")})
    return p


def test_predictor_predict_matches_reference(tmp_path):
    p = _golden_predictor()
    y = p.predict(G["encode/onehot"])
    assert y.shape == (32,) and np.abs(y - G["pred/y"]).max() < 2e-5 * np.abs(G["pred/y"]).max()
    one = p.predict(G["encode/onehot"][:1])
    assert one.shape == () and abs(float(one) - float(G["pred/y_single"])) < 1e-4
    assert np.allclose(p.predict(torch.from_numpy(G["encode/onehot"])), y)
    # the reference's checkpoint format, written and read back
    path = str(tmp_path / "EER.48.Coarse.pth.tar")
    p.save(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    assert set(sd) == {"model", "ynorm"} and "7.bias" in sd["model"] and "running_var" in sd["ynorm"]
    q = nas.AccuracyPredictor(int(G["nfeat/onehot"]), pretrained=path)
    assert np.array_equal(q.predict(G["encode/onehot"]), y)
    with pytest.raises(FileExistsError):
        p.save(path)
    with pytest.raises(FileNotFoundError):
        nas.AccuracyPredictor(48, pretrained=str(tmp_path / "missing.pth.tar"))


@pytest.mark.parametrize("mode", ["Coarse", "Fine"])
def test_minimize_error_encode_matches_reference(mode):
    prob = nas.MinimizeError(None, None, None, space_mode=mode)
    got = prob.encode(G["genome/" + mode])
    want = [[int(v) for v in row if v >= 0] for row in G["genome_encoded/" + mode]]
    assert got == want
    assert prob.encode(G["genome/" + mode][3]) == want[3]            # 1-D genome -> one flat config


def _random_configs(n, seed):
    import random
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        d = rng.choice(CSPACE[0])
        out.append([d] + [rng.choice(CSPACE[1]) for _ in range(d + 1)] + [rng.choice(CSPACE[6]) for _ in range(d + 1)]
                   + [rng.choice(CSPACE[11])])
    return out


def test_predictor_fit_learns_a_synthetic_error_surface():
    codec = _codec()
    X = np.stack([codec.encode(c) for c in _random_configs(1536, seed=1)])
    w = np.random.default_rng(2).normal(size=X.shape[1]).astype(np.float32)
    y = 3.0 + X @ w * 0.3 + 0.2 * (X[:, 3] * X[:, 20])               # EER-like numbers around 3 %
    p = nas.AccuracyPredictor(X.shape[1], seed=4)
    before = float(np.abs(p.predict(X) - y).mean())
    after = p.fit(X, y, n_epoch=40, lr=1e-3)
    Xt = np.stack([codec.encode(c) for c in _random_configs(256, seed=9)])
    yt = 3.0 + Xt @ w * 0.3 + 0.2 * (Xt[:, 3] * Xt[:, 20])
    held_out = float(np.abs(p.predict(Xt) - yt).mean())
    print("L1 error before %.3f after %.3f held-out %.3f (target std %.3f)" % (before, after, held_out, y.std()))
    assert after < 0.15 * y.std() and held_out < 0.3 * y.std() and after < before


class _LinearPredictor:
    def __init__(self, nfeat):
        self.w = np.random.default_rng(11).normal(size=nfeat)

    def predict(self, X):
        return np.asarray(X, np.float64) @ self.w


class _WidthLatency:
    def query(self, subnet):
        depth, widths, kernels, cat = subnet
        return 0.004 * sum(w * k for w, k in zip(widths, kernels)) * depth + cat / 400.0


class _Depth2(nas.GPULatencyConstraint):
    """The coarse space restricted to depth 2 (16 875 subnets): small enough to enumerate."""

    def __init__(self, *a, **k):
        nas.GPULatencyConstraint.__init__(self, *a, **k)
        self.ub = [2, 4, 4, 4, 0, 0, 3, 3, 3, 1, 1, 4]
        self.lb = [2, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 0]


def test_ea_solve_finds_the_exhaustive_optimum():
    codec = _codec()
    prob = _Depth2(codec, _LinearPredictor(codec.nfeat), _WidthLatency(), space_mode="Coarse", threshold=12.0)
    grids = np.meshgrid(*[np.arange(prob.lb[i], prob.ub[i] + 1) for i in range(12)], indexing="ij")
    genomes = np.stack([g.ravel() for g in grids], 1)
    assert len(genomes) == 16875
    obj, bad = prob.evaluate(genomes)
    assert 0.02 < (~bad).mean() < 0.95
    best = obj[~bad].min()
    got = nas.ea_solve("EER", prob, NIND=50, MAXGEN=120, Pm=0.1, verbose=False, seed=3)
    assert got is not None
    subnet, info = got
    assert info["Latency"] < 12.0 and subnet[0] == 2 and len(subnet[1]) == 3 and len(subnet[2]) == 3
    rank = int((obj[~bad] < info["ACC"] - 1e-12).sum())
    print("GA objective %.4f, exhaustive optimum %.4f, better feasible subnets: %d of %d" % (info["ACC"], best, rank, (~bad).sum()))
    assert rank == 0
    # an unsatisfiable constraint reports no solution, as the reference does
    none = _Depth2(codec, _LinearPredictor(codec.nfeat), _WidthLatency(), space_mode="Coarse", threshold=0.1)
    assert nas.ea_solve("EER", none, NIND=20, MAXGEN=5, verbose=False) is None


def test_search_under_the_measured_mi355x_latency_table():
    """The notebook's use (TDNN-NAS.ipynb cells 16-19) with this repository's pieces: the latency table measured on an
    MI355X (profiles/, sugar_amd.TDNNLatency.query), the golden predictor, the whole coarse space.  Tighter latency
    budgets can only cost predicted accuracy, and every answer respects its budget."""
    import sugar_amd
    table = os.path.join(os.path.dirname(os.path.dirname(__file__)), "profiles", "tdnn8m2g_latency_mi355x_80_300.yml")
    lat = sugar_amd.TDNNLatency(table)
    codec = _codec()
    pred = _golden_predictor()
    prev = None
    for threshold in (0.25, 0.40, 0.75):                    # ms for 64 x 300 frames; the table spans 0.18 .. 0.72
        prob = nas.GPULatencyConstraint(codec, pred, lat, space_mode="Coarse", threshold=threshold)
        subnet, info = nas.ea_solve("EER", prob, NIND=40, MAXGEN=60, verbose=False, seed=1)
        assert info["Latency"] < threshold and abs(lat.query(subnet) - info["Latency"]) < 1e-12
        flat = [subnet[0]] + list(subnet[1]) + list(subnet[2]) + [subnet[3]]
        assert abs(float(pred.predict(codec.encode(flat))) - info["ACC"]) < 1e-5
        if prev is not None:
            assert info["ACC"] <= prev + 1e-6               # a looser budget never gives a worse optimum... found by the GA
        prev = info["ACC"]
