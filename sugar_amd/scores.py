"""
Cosine trial scoring with the reference's interface (sugar/scores.py:114-216), computed by one batched HIP
kernel (sugar_cosine_trials) instead of a per-trial python loop on the CPU.
"""
import numpy as np
import torch

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def score_pairs(vec, seg_off, enroll_idx, test_idx):
    """
    vec [n_rows, D] float32 CUDA; seg_off int32 [n_utts+1]; enroll_idx / test_idx int32 [n_trials] (CUDA).
    Returns float32 CUDA scores [n_trials] = mean of the pairwise cosines of each trial's segment vectors.
    """
    if not vec.is_cuda:
        raise _lib.SugarError("sugar_amd runs on an MI355X only: vectors must be CUDA tensors (no CPU fallback)")
    vec = vec.contiguous().float()
    n_trials = int(enroll_idx.numel())
    out = torch.empty(n_trials, dtype=torch.float32, device=vec.device)
    rc = _lib.lib().sugar_cosine_trials(vec.data_ptr(), vec.shape[1], seg_off.data_ptr(), int(seg_off.numel()) - 1,
                                        enroll_idx.data_ptr(), test_idx.data_ptr(), n_trials, out.data_ptr(),
                                        _stream(vec.device))
    _lib.check(rc, "sugar_cosine_trials")
    return out


def calculate_score(enroll_feat, test_feat, mode="cosine"):
    """sugar/scores.py:114-153, cosine mode: mean over the n x m pairwise cosines -> python float."""
    if isinstance(mode, list):
        return [calculate_score(enroll_feat, test_feat, m) for m in mode]
    if mode != "cosine":
        raise ValueError("only mode='cosine' is on the accelerated path (got %r)" % (mode,))
    dev = enroll_feat.device
    vec = torch.cat([enroll_feat, test_feat.to(dev)], dim=0)
    n, m = enroll_feat.shape[0], test_feat.shape[0]
    seg = torch.tensor([0, n, n + m], dtype=torch.int32, device=dev)
    e = torch.zeros(1, dtype=torch.int32, device=dev)
    t = torch.ones(1, dtype=torch.int32, device=dev)
    return float(score_pairs(vec, seg, e, t).item())


def score_trials(trials, vectors, mode="cosine", n_jobs=4):
    """
    sugar/scores.py:156-216.  trials: iterable of {"label", "enroll", "test"} (VerificationTrials, database.py:314-318);
    vectors: {utterance: Tensor[n_seg, D]} as returned by extract_vectors.
    Returns (scores float64 ndarray, labels ndarray, [[enroll, test], ...]) like the reference.
    """
    if mode != "cosine":
        raise ValueError("only mode='cosine' is on the accelerated path (got %r)" % (mode,))
    keys = list(vectors.keys())
    index = {k: i for i, k in enumerate(keys)}
    dev = vectors[keys[0]].device
    seg = [0]
    for k in keys:
        seg.append(seg[-1] + vectors[k].shape[0])
    vec = torch.cat([vectors[k].reshape(-1, vectors[k].shape[-1]) for k in keys], dim=0)
    labels, pairs, e_idx, t_idx = [], [], [], []
    for tr in trials:
        labels.append(tr["label"])
        pairs.append([tr["enroll"], tr["test"]])
        e_idx.append(index[tr["enroll"]])
        t_idx.append(index[tr["test"]])
    scores = score_pairs(vec, torch.tensor(seg, dtype=torch.int32, device=dev),
                         torch.tensor(e_idx, dtype=torch.int32, device=dev),
                         torch.tensor(t_idx, dtype=torch.int32, device=dev))
    return scores.double().cpu().numpy(), np.array(labels), pairs
