"""
Verification metrics of the evaluation call (`veri_validate`, sugar/models/__init__.py:535-595): EER and minDCF.

Host-side numpy on a score array that already left the GPU (37 720 floats for Vox1-O); same names, arguments and
return values as sugar/metrics.py:45-218.  The reference's `calculate_eer` ends in `np.float(thresh)`, which raises
on numpy >= 1.24; here it returns a python float.
"""
import numpy as np


def _roc_curve(y, y_score, pos=1):
    """ROC points as sklearn.metrics.roc_curve(drop_intermediate=False) returns them: thresholds descending,
    first point (0, 0) at threshold +inf, one point per distinct score."""
    y = np.asarray(y) == pos
    y_score = np.asarray(y_score, dtype=np.float64)
    order = np.argsort(-y_score, kind="mergesort")
    y, y_score = y[order], y_score[order]
    last = np.r_[np.where(np.diff(y_score))[0], y.size - 1]      # last index of every run of equal scores
    tps = np.cumsum(y)[last].astype(np.float64)
    fps = (1 + last - tps).astype(np.float64)
    tps = np.r_[0.0, tps]
    fps = np.r_[0.0, fps]
    thr = np.r_[np.inf, y_score[last]]
    return fps / fps[-1], tps / tps[-1], thr


def calculate_eer(y, y_score, pos=1):
    """
    Equal error rate (sugar/metrics.py:45-74): the root of 1 - x - TPR(x) on the piecewise-linear ROC curve, and the
    score threshold interpolated at that false-positive rate.  2-D scores: the column with the lowest EER.
    Returns (eer, thresh).
    """
    y_score = np.asarray(y_score)
    assert y_score.ndim == 1 or y_score.ndim == 2
    if y_score.ndim == 2:
        both = np.array([calculate_eer(y, y_score[:, i], pos) for i in range(y_score.shape[1])])
        best = both[:, 0].argmin()
        return both[best, 0], float(both[best, 1])
    fpr, tpr, thr = _roc_curve(y, y_score, pos)
    # g(x) = 1 - x - tpr(x) is continuous, piecewise linear and non-increasing from g(0) >= 0 to g(1) = -1
    g = 1.0 - fpr - tpr
    k = int(np.argmax(g <= 0.0))                                   # first vertex at or below zero
    if g[k] == 0.0 or k == 0:
        eer, t = fpr[k], thr[k]
    else:
        w = g[k - 1] / (g[k - 1] - g[k])                            # zero crossing inside segment (k-1, k)
        eer = fpr[k - 1] + w * (fpr[k] - fpr[k - 1])
        # threshold: linear interpolation over fpr, as interp1d(fpr, thresholds)(eer) in the reference; a vertical
        # ROC segment (equal fpr) takes the later vertex, and the +inf of vertex 0 is replaced like sklearn < 1.3
        # did (max score + 1) so the interpolation stays finite
        t0 = thr[k - 1] if np.isfinite(thr[k - 1]) else thr[1] + 1.0
        t = thr[k] if fpr[k] == fpr[k - 1] else t0 + (eer - fpr[k - 1]) / (fpr[k] - fpr[k - 1]) * (thr[k] - t0)
    return float(eer), float(t)


def compute_error_rates(scores, labels):
    """sugar/metrics.py:77-135 (Kaldi compute_min_dcf.py): fnr / fpr at every score taken as the threshold,
    scores ascending."""
    scores = np.asarray(scores, dtype=np.float64)
    labels = np.asarray(labels)
    order = np.argsort(scores)
    thresholds = scores[order]
    labels = labels[order].astype(np.float64)
    fn = np.cumsum(labels)
    tn = np.cumsum(1.0 - labels)
    fnrs = fn / float(labels.sum())
    fprs = 1.0 - tn / float(labels.size - labels.sum())
    return fnrs, fprs, thresholds


def compute_mindcf(fnrs, fprs, thresholds, p_target=0.01, c_miss=1, c_fa=1, mode="vectorized"):
    """sugar/metrics.py:138-177: min over thresholds of c_miss fnr p + c_fa fpr (1 - p), normalised (NIST SRE16)."""
    fnrs, fprs = np.asarray(fnrs), np.asarray(fprs)
    c_det = c_miss * fnrs * p_target + c_fa * fprs * (1 - p_target)
    i = int(c_det.argmin())
    c_def = min(c_miss * p_target, c_fa * (1 - p_target))
    return c_det[i] / c_def, thresholds[i]


def calculate_mindcf(y, y_score, p_target=0.01, c_miss=1, c_fa=1):
    """sugar/metrics.py:180-218.  Returns (mindcf, threshold); 2-D scores: the column with the lowest minDCF."""
    y_score = np.asarray(y_score)
    assert y_score.ndim == 1 or y_score.ndim == 2
    if y_score.ndim == 2:
        both = np.array([calculate_mindcf(y, y_score[:, i], p_target, c_miss, c_fa) for i in range(y_score.shape[1])])
        best = both[:, 0].argmin()
        return both[best, 0], both[best, 1]
    fnrs, fprs, thresholds = compute_error_rates(y_score, y)
    return compute_mindcf(fnrs, fprs, thresholds, p_target=p_target, c_miss=c_miss, c_fa=c_fa)


def calculate_val(y, y_score, far=1e-3):
    """
    Validation rate at a false-accept rate (sugar/metrics.py:221-281): bisection on the threshold until the interval is
    below far * 1e-3, VAL = accepted targets / targets at the upper end; a 2-d score array gives the column with the
    SMALLEST VAL (`argmin`, as written there).  Returns (VAL, threshold).
    """
    y, y_score = np.asarray(y), np.asarray(y_score)
    assert y_score.ndim == 1 or y_score.ndim == 2
    if y_score.ndim == 2:
        vt = np.array([calculate_val(y, y_score[:, i], far=far) for i in range(y_score.shape[1])])
        best = vt[:, 0].argmin()
        return vt[best, 0], vt[best, 1]
    lo, hi = y_score.min(), y_score.max()
    nontar = y == 0
    n_nontar = nontar.sum()
    while (hi - lo) >= far * 1e-3:
        mid = (lo + hi) / 2
        if (nontar & (y_score >= mid)).sum() / n_nontar > far:
            lo = mid
        else:
            hi = mid
    assert (nontar & (y_score >= hi)).sum() / n_nontar <= far
    target = y == 1
    return y[target & (y_score >= hi)].sum() / target.sum(), hi
