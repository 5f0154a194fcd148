"""
Trial scoring with the reference's interface (sugar/scores.py:114-216): modes "cosine", "norm2" (the reference's
default for calculate_score) and lists of modes, each computed by one batched HIP kernel (sugar_cosine_trials,
sugar_norm2_trials) instead of a per-trial python loop on the CPU.
"""
import numpy as np
import torch

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


MODES = ("cosine", "norm2")


def _check_mode(mode):
    """A mode string or a list of them (sugar/scores.py:140, 151); anything else is an error (the reference falls
    through to an UnboundLocalError there)."""
    for m in (mode if isinstance(mode, list) else [mode]):
        if m not in MODES:
            raise ValueError("scoring mode %r not in %s" % (m, list(MODES)))


def score_pairs(vec, seg_off, enroll_idx, test_idx, mode="cosine"):
    """
    vec [n_rows, D] float32 CUDA; seg_off int32 [n_utts+1]; enroll_idx / test_idx int32 [n_trials] (CUDA).
    Returns float32 CUDA scores [n_trials]: "cosine" = mean of the pairwise cosines of each trial's segment vectors
    (sugar/scores.py:146-150); "norm2" = minus the mean pairwise_distance as the reference calls it (:141-145; the
    2-norm runs over the test segments, see include/sugar_amd.h).
    """
    if not vec.is_cuda:
        raise _lib.SugarError("sugar_amd runs on an MI355X only: vectors must be CUDA tensors (no CPU fallback)")
    _check_mode(mode)
    vec = vec.contiguous().float()
    n_trials = int(enroll_idx.numel())
    out = torch.empty(n_trials, dtype=torch.float32, device=vec.device)
    fn = _lib.lib().sugar_cosine_trials if mode == "cosine" else _lib.lib().sugar_norm2_trials
    rc = fn(vec.data_ptr(), vec.shape[1], seg_off.data_ptr(), int(seg_off.numel()) - 1, enroll_idx.data_ptr(),
            test_idx.data_ptr(), n_trials, out.data_ptr(), _stream(vec.device))
    _lib.check(rc, "sugar_%s_trials" % mode)
    return out


def calculate_score(enroll_feat, test_feat, mode="norm2"):
    """sugar/scores.py:114-153: "cosine" = mean over the n x m pairwise cosines, "norm2" (the reference's default) =
    minus the mean pairwise distance -> python float; a list of modes -> list of floats."""
    _check_mode(mode)
    if isinstance(mode, list):
        return [calculate_score(enroll_feat, test_feat, m) for m in mode]
    dev = enroll_feat.device
    vec = torch.cat([enroll_feat, test_feat.to(dev)], dim=0)
    n, m = enroll_feat.shape[0], test_feat.shape[0]
    seg = torch.tensor([0, n, n + m], dtype=torch.int32, device=dev)
    e = torch.zeros(1, dtype=torch.int32, device=dev)
    t = torch.ones(1, dtype=torch.int32, device=dev)
    return float(score_pairs(vec, seg, e, t, mode).item())


def score_trials(trials, vectors, mode="cosine", n_jobs=4):
    """
    sugar/scores.py:156-216.  trials: iterable of {"label", "enroll", "test"} (VerificationTrials, database.py:314-318);
    vectors: {utterance: Tensor[n_seg, D]} as returned by extract_vectors.
    mode: "cosine", "norm2" or a list of them (then scores is [n_trials, len(mode)], one column per mode, :175-177).
    Returns (scores float64 ndarray, labels ndarray, [[enroll, test], ...]) like the reference.
    """
    scores, labels, pairs = score_trials_device(trials, vectors, mode)
    return scores.double().cpu().numpy(), labels, pairs


def score_trials_device(trials, vectors, mode="cosine"):
    """score_trials up to the copy to the host: (scores on the device, float32 [n_trials] or [n_trials, len(mode)]; labels
    ndarray; [[enroll, test], ...]).  Enqueued on the current stream and not waited for, so a caller (the NAS sweep) can
    start the next model's extraction before it reads these scores back."""
    _check_mode(mode)
    matrix = getattr(vectors, "matrix", None)
    if matrix is not None:                       # a VectorTable from extract_vectors: one matrix, offsets on the device
        keys, vec, seg_off, dev = vectors.files, matrix, vectors.seg_off, matrix.device
    else:
        keys = list(vectors.keys())
        dev = vectors[keys[0]].device
        seg = [0]
        for k in keys:
            seg.append(seg[-1] + vectors[k].shape[0])
        vec = torch.cat([vectors[k].reshape(-1, vectors[k].shape[-1]) for k in keys], dim=0)
        seg_off = torch.tensor(seg, dtype=torch.int32, device=dev)
    labels, pairs, e_dev, t_dev = _trial_index(trials, keys, dev)
    if isinstance(mode, list):
        scores = torch.stack([score_pairs(vec, seg_off, e_dev, t_dev, m) for m in mode], dim=1)
    else:
        scores = score_pairs(vec, seg_off, e_dev, t_dev, mode)
    return scores, labels, pairs


def _fingerprint(trials):
    """Cheap content check of a VerificationTrials-shaped list: its length and 32 sampled (label, enroll, test)."""
    n = len(trials.labels)
    step = max(1, n // 32)
    return (n, tuple((trials.labels[i], trials.enrolls[i], trials.tests[i]) for i in range(0, n, step)))


def _trial_index(trials, keys, dev):
    """(labels ndarray, [[enroll, test], ...], enroll row index, test row index on `dev`) of a trial list against the
    utterance order `keys`.  A sweep scores the same list once per subnet, so the result is remembered -- only on
    VerificationTrials-shaped objects (`labels` / `enrolls` / `tests` lists filled once by load_trial), keyed by the
    file list (the same object, or an equal one: every extract_vectors call over a file list builds a new list of the
    same names, and re-indexing 37 720 trials is 60 ms of interpreter time), the device and a sampled content
    fingerprint; plain lists are indexed every time."""
    memo_ok = all(isinstance(getattr(trials, a, None), list) for a in ("labels", "enrolls", "tests"))
    fp = _fingerprint(trials) if memo_ok else None
    memo = getattr(trials, "_sugar_trial_index", None) if memo_ok else None
    if memo is not None and (memo[0] is keys or memo[0] == keys) and memo[1] == fp and memo[2] == dev:
        return memo[3]
    index = {k: i for i, k in enumerate(keys)}
    labels, pairs, e_idx, t_idx = [], [], [], []
    for tr in trials:
        labels.append(tr["label"])
        pairs.append([tr["enroll"], tr["test"]])
        e_idx.append(index[tr["enroll"]])
        t_idx.append(index[tr["test"]])
    out = (np.array(labels), pairs, torch.from_numpy(np.asarray(e_idx, dtype=np.int32)).to(dev),
           torch.from_numpy(np.asarray(t_idx, dtype=np.int32)).to(dev))
    if memo_ok:
        trials._sugar_trial_index = (keys, fp, dev, out)      # `keys is` the packed list's file list
    return out


# ------------------------------------------------------------------------------------------------ score normalisation
class CohortScores(dict):
    """
    What score_cohorts returns: {utterance: float32 ndarray [n_cohort]} like the reference (sugar/scores.py:96-111),
    materialised lazily -- the [n_utts, n_cohort] matrix stays on the GPU (`.matrix`, rows in `.keys_order`) so that
    asnorm / snorm run there without a 117 MB round trip; the first dict access copies it to the host once.
    """

    def __init__(self, keys, matrix):
        super().__init__()
        self.keys_order = list(keys)
        self.matrix = matrix
        self._host = False

    def _materialise(self):
        if not self._host:
            self._host = True
            host = self.matrix.cpu().numpy()
            for i, k in enumerate(self.keys_order):
                dict.__setitem__(self, k, host[i])

    def __getitem__(self, k):
        self._materialise()
        return dict.__getitem__(self, k)

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def keys(self):
        return list(self.keys_order)

    def __iter__(self):
        return iter(self.keys_order)

    def __len__(self):
        return len(self.keys_order)

    def __contains__(self, k):
        self._materialise()
        return dict.__contains__(self, k)


def score_cohorts(cohorts, vectors, mode="cosine", device="cuda:0"):
    """
    sugar/scores.py:77-111: for every utterance the mean (over its segment vectors) cosine similarity with each of the
    cohort vectors.  cohorts: Tensor [n_cohort, D]; vectors: {utterance: Tensor [n_seg, D]}.
    Returns a CohortScores dict {utterance: ndarray [n_cohort]}.
    """
    if mode != "cosine":
        raise ValueError("only mode='cosine' is supported (as in the reference)")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.SugarError("sugar_amd runs on an MI355X only: pass device='cuda:N' (there is no CPU fallback)")
    matrix = getattr(vectors, "matrix", None)
    if matrix is not None and matrix.device == dev and len(vectors.files) == len(vectors):
        # a VectorTable from extract_vectors (no file listed twice): its matrix and offsets as they are, instead of
        # 4 874 views concatenated again
        keys, vec, seg_off = list(vectors.files), matrix.float().contiguous(), vectors.seg_off
    else:
        keys = list(vectors.keys())
        seg = [0]
        for k in keys:
            seg.append(seg[-1] + vectors[k].reshape(-1, vectors[k].shape[-1]).shape[0])
        vec = torch.cat([vectors[k].reshape(-1, vectors[k].shape[-1]).to(dev) for k in keys], dim=0).float().contiguous()
        seg_off = torch.tensor(seg, dtype=torch.int32, device=dev)
    coh = cohorts.to(dev).float().contiguous()
    n_utts, n_coh, dim = len(keys), coh.shape[0], vec.shape[1]
    if coh.shape[1] != dim:
        raise ValueError("cohort vectors have %d dimensions, utterance vectors %d" % (coh.shape[1], dim))
    scratch = torch.empty((n_utts + n_coh) * dim, dtype=torch.float32, device=dev)
    out = torch.empty(n_utts, n_coh, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().sugar_cohort_scores(vec.data_ptr(), dim, seg_off.data_ptr(), n_utts, coh.data_ptr(), n_coh,
                                             scratch.data_ptr(), out.data_ptr(), _stream(dev)), "sugar_cohort_scores")
    return CohortScores(keys, out)


def _cohort_matrix(cohort, device):
    if isinstance(cohort, CohortScores):
        return cohort.keys_order, cohort.matrix
    keys = list(cohort.keys())
    mat = torch.from_numpy(np.stack([np.asarray(cohort[k], dtype=np.float32) for k in keys])).to(device)
    return keys, mat.contiguous()


def _trial_columns(scores):
    """scores: DataFrame (or mapping) with columns score / enroll / test, as in the reference."""
    return (np.asarray(scores["score"], dtype=np.float64), list(scores["enroll"]), list(scores["test"]))


def _snorm_gpu(scores, cohort, ntop, divisor, std_eps, device):
    orig, enr, tst = _trial_columns(scores)
    dev = torch.device(device)
    keys, mat = _cohort_matrix(cohort, dev)
    index = {k: i for i, k in enumerate(keys)}
    n_rows, n_cols = mat.shape
    mean = torch.empty(n_rows, dtype=torch.float32, device=dev)
    std = torch.empty(n_rows, dtype=torch.float32, device=dev)
    st = _stream(dev)
    _lib.check(_lib.lib().sugar_topk_stats(mat.data_ptr(), n_rows, n_cols, int(ntop), float(divisor), mean.data_ptr(),
                                          std.data_ptr(), st), "sugar_topk_stats")
    s = torch.from_numpy(orig.astype(np.float32)).to(dev)
    e = torch.tensor([index[k] for k in enr], dtype=torch.int32, device=dev)
    t = torch.tensor([index[k] for k in tst], dtype=torch.int32, device=dev)
    out = torch.empty(len(orig), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().sugar_snorm_trials(s.data_ptr(), e.data_ptr(), t.data_ptr(), len(orig), mean.data_ptr(),
                                            std.data_ptr(), float(std_eps), out.data_ptr(), st), "sugar_snorm_trials")
    return out.double().cpu().numpy()


def asnorm(scores, cohort, nTop=300, std_eps=1e-8, device="cuda:0"):
    """
    Adaptive s-norm (sugar/scores.py:332-400): per utterance, mean / std of its nTop most similar cohort scores
    (nTop < 0: the |nTop| least similar); every trial score becomes the average of its z-norm (enrol statistics)
    and t-norm (test statistics).  scores: DataFrame with columns score / enroll / test; cohort: score_cohorts'
    return value (or any {utterance: ndarray}).  Returns a float64 ndarray.
    """
    return _snorm_gpu(scores, cohort, nTop, abs(nTop), std_eps, device)


def snorm(scores, cohort, nTop=None, std_eps=1e-8, device="cuda:0"):
    """
    S-norm (sugar/scores.py:269-329).  nTop given -> asnorm.  Otherwise all cohort scores are used and the standard
    deviation is sqrt(sum((v - mean)^2) / N) with N = the number of TRIALS, exactly as the reference computes it
    (:295, :308).
    """
    if nTop is not None:
        return asnorm(scores, cohort, nTop, std_eps, device)
    n_trials = len(_trial_columns(scores)[0])
    keys, mat = _cohort_matrix(cohort, torch.device(device))
    return _snorm_gpu(scores, cohort, mat.shape[1], n_trials, std_eps, device)


def score_batch(vec, batch, mode="cosine", device="cuda:0"):
    """
    sugar/scores.py:56-74: cosine similarity of every row of `vec` [n, D] with every row of `batch` [m, D] -> float32
    ndarray [n, m] (F.cosine_similarity broadcast, eps 1e-8).  One sugar_cohort_scores launch with every row of `vec` as
    an utterance of its own.
    """
    if mode != "cosine":
        raise ValueError("only mode='cosine' is supported (as in the reference)")
    dev = vec.device if vec.is_cuda else torch.device(device)
    if dev.type != "cuda":
        raise _lib.SugarError("sugar_amd runs on an MI355X only: pass device='cuda:N' (there is no CPU fallback)")
    v = vec.reshape(-1, vec.shape[-1]).to(dev).float().contiguous()
    b = batch.reshape(-1, batch.shape[-1]).to(dev).float().contiguous()
    n, m, dim = v.shape[0], b.shape[0], v.shape[1]
    seg_off = torch.arange(n + 1, dtype=torch.int32, device=dev)
    scratch = torch.empty((n + m) * dim, dtype=torch.float32, device=dev)
    out = torch.empty(n, m, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().sugar_cohort_scores(v.data_ptr(), dim, seg_off.data_ptr(), n, b.data_ptr(), m, scratch.data_ptr(),
                                             out.data_ptr(), _stream(dev)), "sugar_cohort_scores")
    return out.cpu().numpy()


def ztnorm(scores, cohort, nTop=None):
    """
    ZT-norm (sugar/scores.py:219-266), host-side as in the reference: per utterance the mean / population std of its
    cohort scores -- with nTop the ascending-sorted scores WITHOUT their first nTop entries (nTop > 0: `value[nTop:]`) or
    only the first |nTop| (nTop < 0: `value[:-nTop]`), exactly as written there -- then z-norm with the enrol statistics
    followed by t-norm with the test statistics, each skipped where the std is 0.  scores: DataFrame / mapping with
    columns score / enroll / test; cohort: score_cohorts' return value or any {utterance: ndarray}.  Returns float64.
    """
    orig, enr, tst = _trial_columns(scores)
    mean, std = {}, {}
    for key in set(enr) | set(tst):
        value = np.asarray(cohort[key])
        if nTop is not None:
            value = np.sort(value, kind="mergesort")
            value = value[:-nTop] if nTop < 0 else value[nTop:]
        mean[key], std[key] = value.mean(), value.std()
    out = np.empty(len(orig), dtype=np.float64)
    for i, (s, e, t) in enumerate(zip(orig, enr, tst)):
        z = (s - mean[e]) / std[e] if std[e] != 0 else s
        out[i] = (z - mean[t]) / std[t] if std[t] != 0 else z
    return out


def normalize_scores(scores, s_min=0, s_max=1):
    """sugar/scores.py:402-420 (sklearn's minmax_scale): scores mapped linearly onto [s_min, s_max]."""
    x = np.asarray(scores, dtype=np.float64)
    lo, hi = x.min(axis=0), x.max(axis=0)
    span = np.where(hi - lo == 0, 1.0, hi - lo)               # sklearn: a constant column keeps scale 1
    return (x - lo) / span * (s_max - s_min) + s_min


def save_sctr(sc, tr, score_file, root_dir="", scale=True):
    """sugar/scores.py:27-36: one "%.6f enroll test" line per trial (scores min-max scaled to [0, 1] when `scale`),
    `root_dir` stripped from both names; refuses to overwrite."""
    import os
    assert not os.path.exists(score_file)
    if scale is True:
        sc = normalize_scores(sc)
    with open(score_file, "w") as f:
        for s, t in zip(sc, tr):
            f.write("%.6f %s %s\n" % (s, t[0].replace(root_dir, ""), t[1].replace(root_dir, "")))


def load_sctr(score_file):
    """sugar/scores.py:38-53: (scores, [[enroll, test], ...]) of a file written by save_sctr."""
    sc, tr = [], []
    with open(score_file, "r") as f:
        for line in f:
            s, t1, t2 = line.split(" ")
            sc.append(float(s))
            tr.append([t1, t2.replace("\n", "")])
    return sc, tr
