"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (sugar_amd/).

A plain-torch / numpy CPU restatement of the reference's embedding-extraction path
(mechanicalsea/sugar, EfficientTDNN).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file, and only as the checker.

Pinning status
--------------
* Backbone (weight materialisation, clone, forward), L2-norm and cosine scoring are
  PINNED: oracle/make_golden.py imports the real reference classes from /root/reference
  in the build container, runs them on the deterministic weights of sugar_amd/synth.py
  and commits the outputs to tests/golden/; tests/test_oracle.py checks this file against
  those vectors.
* Front-end (LogMelFbanks) is PARITY UNPINNED against the reference: its arithmetic lives in
  torchaudio (``torchaudio.transforms.MelSpectrogram``, sugar/transforms.py:50-52), which is
  neither vendored in /root/reference, nor version-pinned there, nor installed here, and the
  reference holds no fixture at that boundary.  This file restates torchaudio's published
  algorithm (Spectrogram -> torch.stft(center=True, pad_mode='reflect', window zero-padded to
  n_fft) -> |.|^2 -> MelScale with HTK triangular filters, norm=None) and cross-checks it with an
  independent float64 numpy framing + rFFT (``logmel_fbank_np64``).

All file:line citations are relative to /root/reference/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
N_FFT = 512
WIN_LENGTH = 400
HOP = 160
F_MIN = 20.0
F_MAX = 7600.0
N_MELS = 80
BN_EPS = 1e-5
SCALE = 8          # Res2 scale, dynamictdnn.py:384
SE_REDUCTION = 4   # dynamictdnn.py:82
BOTTLENECK = 128   # dynamictdnn.py:494


# --------------------------------------------------------------------------- front-end
def num_frames(num_samples: int) -> int:
    """center=True STFT: T = 1 + floor(L / hop)."""
    return 1 + num_samples // HOP


def preemphasis(x: torch.Tensor, coef: float = 0.97) -> torch.Tensor:
    """sugar/transforms.py:22-27: reflect-pad one sample on the left, then y[t] = x[t] - coef * x[t-1]."""
    assert x.dim() == 2
    prev = torch.cat([x[:, 1:2], x[:, :-1]], dim=1)  # reflect pad (1, 0): x[-1] := x[1]
    return x - coef * prev


def hamming_window_periodic(n: int = WIN_LENGTH, dtype=torch.float32) -> torch.Tensor:
    """torch.hamming_window(n) default periodic=True: 0.54 - 0.46 cos(2 pi k / n)."""
    k = torch.arange(n, dtype=torch.float64)
    return (0.54 - 0.46 * torch.cos(2.0 * math.pi * k / n)).to(dtype)


def mel_filterbank(n_freqs: int = N_FFT // 2 + 1, f_min: float = F_MIN, f_max: float = F_MAX,
                   n_mels: int = N_MELS, sample_rate: int = SAMPLE_RATE) -> torch.Tensor:
    """
    torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> [n_freqs, n_mels] float32
    (the MelScale used by MelSpectrogram at sugar/transforms.py:50-52).
    """
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def mel_power(x: torch.Tensor) -> torch.Tensor:
    """PreEmphasis + MelSpectrogram (sugar/transforms.py:48-53): [B, L] -> [B, 80, T] mel power."""
    y = preemphasis(x.float())
    spec = torch.stft(y, n_fft=N_FFT, hop_length=HOP, win_length=WIN_LENGTH,
                      window=hamming_window_periodic(), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    power = spec.abs().pow(2.0)                       # [B, 257, T]
    fb = mel_filterbank()
    return torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)


def logmel_fbank(x: torch.Tensor) -> torch.Tensor:
    """LogMelFbanks.forward (sugar/transforms.py:60-77) without the masker branch: [B, L] -> [B, 1, 80, T]."""
    with torch.no_grad():
        m = (mel_power(x) + 1e-6).log()
        m = F.instance_norm(m, eps=1e-5)              # InstanceNorm1d(80): biased var over T, no affine (:47,73)
        return m.unsqueeze(1)


def logmel_fbank_np64(x: np.ndarray) -> np.ndarray:
    """Independent float64 restatement (explicit reflect padding + framing + np.fft.rfft): [L] -> [80, T]."""
    x = np.asarray(x, dtype=np.float64)
    L = x.shape[0]
    y = x - 0.97 * np.concatenate([x[1:2], x[:-1]])
    pad = N_FFT // 2
    yp = np.concatenate([y[1:pad + 1][::-1], y, y[-pad - 1:-1][::-1]])
    T = num_frames(L)
    win = np.zeros(N_FFT)
    k = np.arange(WIN_LENGTH)
    left = (N_FFT - WIN_LENGTH) // 2
    win[left:left + WIN_LENGTH] = 0.54 - 0.46 * np.cos(2.0 * np.pi * k / WIN_LENGTH)
    frames = np.stack([yp[t * HOP:t * HOP + N_FFT] * win for t in range(T)])
    power = np.abs(np.fft.rfft(frames, axis=1)) ** 2   # [T, 257]
    mel = power @ mel_filterbank().double().numpy()    # [T, 80]
    lm = np.log(mel + 1e-6).T                          # [80, T]
    mu = lm.mean(axis=1, keepdims=True)
    var = lm.var(axis=1, keepdims=True)
    return (lm - mu) / np.sqrt(var + 1e-5)


# --------------------------------------------------------------------------- weight materialisation
def conv_active_weight(w, in_sel, out_sel, k_active, k53=None, k31=None):
    """
    DynamicConv1d.weight (sugar/modules/dynamicarch.py:411-432).
    in_sel / out_sel: int (top-k slice) or LongTensor (index_select); 5->3 and 3->1 kernel shrink is
    centre crop ``[:, :, 1:-1]`` followed by matmul with the learned k53 / k31.
    """
    w = w[:, :in_sel] if isinstance(in_sel, int) else w.index_select(1, in_sel)
    w = w[:out_sel] if isinstance(out_sel, int) else w.index_select(0, out_sel)
    k_full = w.shape[2]
    assert k_full in (1, 3, 5), "supernet kernels beyond 5 are not on the tdnn8m2g path"
    if k_full > 3 and k_active < 5:
        w = torch.matmul(w[:, :, 1:-1], k53)
    if k_full > 1 and k_active < 3:
        w = torch.matmul(w[:, :, 1:-1], k31)
    return w.contiguous()


def _sel(t, sel):
    return t[:sel] if isinstance(sel, int) else t.index_select(0, sel)


def clone_state(sup: dict, config, num_blocks: int = 4, channels: int = 512, catconv: int = 1536) -> dict:
    """
    DynamicTDNN.clone(config) restated on a *supernet* backbone state dict (keys without the
    ``module.__S__.`` prefix): returns the cloned-subnet state dict (keys as SURVEY Appendix A).
    Follows dynamictdnn.py:668-679 (_module_configs), :409-417 (block index configs),
    :713-730 (clone) and dynamicarch.py:123-142, :265-291, :411-432.
    """
    depth, chans, kernels, ccat = config
    c0 = chans[0]
    out = {}

    def bn(dst, src, sel):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            out["%s.%s" % (dst, leaf)] = _sel(sup["%s.bn.%s" % (src, leaf)], sel).clone()
        out[dst + ".num_batches_tracked"] = sup[src + ".bn.num_batches_tracked"].clone()

    def conv(dst, src, in_sel, out_sel, k_active):
        out[dst + ".weight"] = conv_active_weight(sup[src + ".conv.weight"], in_sel, out_sel, k_active,
                                                  sup.get(src + ".k53"), sup.get(src + ".k31"))

    # layer1: (in_feats, C0, K0) top-k                                        dynamictdnn.py:670
    conv("layer1.conv", "layer1.conv", 80, c0, kernels[0])
    bn("layer1.bn", "layer1.bn", c0)
    for i in range(depth):
        ci, ki = chans[i + 1], kernels[i + 1]
        w = ci // SCALE                                                     # dynamictdnn.py:410
        mid_idx = torch.arange(channels).view(SCALE, -1)[:, :w].flatten().long()  # :411
        p = "blocks.%d." % i
        conv(p + "first_conv.conv", p + "first_conv.conv", c0, mid_idx, 1)
        bn(p + "first_conv.bn", p + "first_conv.bn", mid_idx)
        for j in range(SCALE - 1):
            q = p + "resconv.convs.%d." % j
            conv(q + "conv", q + "conv", w, w, ki)
            bn(q + "bn", q + "bn", w)
        conv(p + "last_conv.conv", p + "last_conv.conv", mid_idx, c0, 1)
        bn(p + "last_conv.bn", p + "last_conv.bn", c0)
        r = c0 // SE_REDUCTION                                              # dynamictdnn.py:105-106
        out[p + "se.reduce.weight"] = sup[p + "se.reduce.linear.weight"][:r, :c0].clone()
        out[p + "se.reduce.bias"] = sup[p + "se.reduce.linear.bias"][:r].clone()
        out[p + "se.expand.weight"] = sup[p + "se.expand.linear.weight"][:c0, :r].clone()
        out[p + "se.expand.bias"] = sup[p + "se.expand.linear.bias"][:c0].clone()
    cat_idx = torch.arange(channels * num_blocks).view(num_blocks, -1)[:depth, :c0].flatten().long()  # :672
    conv("catconv", "catconv", cat_idx, ccat, 1)
    conv("pooling.linear1", "pooling.linear1", ccat, BOTTLENECK, 1)
    conv("pooling.linear2", "pooling.linear2", BOTTLENECK, ccat, 1)
    pool_idx = torch.arange(catconv * 2).view(2, -1)[:, :ccat].flatten().long()                       # :675
    bn("bnpool", "bnpool", pool_idx)
    out["linear.weight"] = sup["linear.linear.weight"].index_select(1, pool_idx).clone()               # :677
    out["linear.bias"] = sup["linear.linear.bias"].clone()
    for leaf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
        out["bnembed." + leaf] = sup["bnembed." + leaf].clone()
    return out


# --------------------------------------------------------------------------- backbone forward
_TRAIN_MOMENTUM = None      # set by subnet_forward(train_momentum=...): train-mode BN, running stats updated in place


def _bn(x, sd, prefix):
    """Eval-mode BN, or -- inside subnet_forward(train_momentum=m) -- nn.BatchNorm1d in train(): batch statistics,
    running_mean / running_var of `sd` updated in place (DynamicBatchNorm1d.forward, dynamicarch.py:254-263)."""
    train = _TRAIN_MOMENTUM is not None
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], train, _TRAIN_MOMENTUM if train else 0.0, BN_EPS)


def _conv(x, w, dilation=1):
    k = w.shape[2]
    return F.conv1d(x, w, None, 1, ((k - 1) // 2) * dilation, dilation)   # dynamicarch.py:377,408-409


def _conv_relu_bn(x, sd, prefix, dilation=1):
    """DynamicConv1dReLUBN.forward: bn(relu(conv(x))) -- conv -> ReLU -> BN (dynamictdnn.py:183)."""
    return _bn(F.relu(_conv(x, sd[prefix + ".conv.weight"], dilation)), sd, prefix + ".bn")


def subnet_forward(sd: dict, depth: int, feats: torch.Tensor, stages: dict = None,
                   train_momentum: float = None) -> torch.Tensor:
    """
    Cloned-subnet forward (DynamicTDNN.forward, dynamictdnn.py:645-666) on a cloned state dict.
    feats [B, 80, T] -> [B, 192].  If ``stages`` is a dict, per-stage activations are stored in it.
    ``train_momentum``: run as one batch_forward step (sugar/models/__init__.py:230-238: model.train(); forward) --
    every BN uses this batch's statistics and updates the running statistics held in ``sd`` IN PLACE.
    """
    global _TRAIN_MOMENTUM
    _TRAIN_MOMENTUM = train_momentum
    try:
        return _subnet_forward(sd, depth, feats, stages)
    finally:
        _TRAIN_MOMENTUM = None


def _subnet_forward(sd, depth, feats, stages):
    def keep(name, t):
        if stages is not None:
            stages[name] = t.clone()

    with torch.no_grad():
        out = _conv_relu_bn(feats, sd, "layer1")
        keep("layer1", out)
        cat_out = []
        sum_out = out
        for i in range(depth):
            p = "blocks.%d." % i
            dil = i + 2                                                   # dynamictdnn.py:629-632
            h = _conv_relu_bn(out, sd, p + "first_conv")
            keep("b%d.first" % i, h)
            w = h.shape[1] // SCALE
            spx = torch.split(h, w, 1)                                    # dynamictdnn.py:291-303
            outs = []
            sp = None
            for j in range(SCALE - 1):
                q = p + "resconv.convs.%d" % j
                inp = spx[j] if j == 0 else spx[j] + sp
                sp = _bn(F.relu(_conv(inp, sd[q + ".conv.weight"], dil)), sd, q + ".bn")
                outs.append(sp)
            outs.append(spx[SCALE - 1])
            h = torch.cat(outs, dim=1)
            keep("b%d.res2" % i, h)
            h = _conv_relu_bn(h, sd, p + "last_conv")
            keep("b%d.last" % i, h)
            s = h.mean(dim=2)                                             # SE, dynamictdnn.py:94-100
            s = F.relu(F.linear(s, sd[p + "se.reduce.weight"], sd[p + "se.reduce.bias"]))
            s = torch.sigmoid(F.linear(s, sd[p + "se.expand.weight"], sd[p + "se.expand.bias"]))
            h = h * s.unsqueeze(2)
            out = h + sum_out                                             # dynamictdnn.py:657-660
            keep("b%d.out" % i, out)
            cat_out.append(out)
            sum_out = sum_out + out
        h = torch.cat(cat_out, dim=1)
        h = F.relu(_conv(h, sd["catconv.weight"]))                        # :663, no BN
        keep("catconv", h)
        a = torch.tanh(_conv(h, sd["pooling.linear1.weight"]))            # dynamictdnn.py:505-515
        a = torch.softmax(_conv(a, sd["pooling.linear2.weight"]), dim=2)
        mean = torch.sum(a * h, dim=2)
        resid = torch.sum(a * h ** 2, dim=2) - mean ** 2
        std = torch.sqrt(resid.clamp(min=1e-9))
        pooled = torch.cat([mean, std], dim=1)
        keep("pool", pooled)
        z = _bn(pooled, sd, "bnpool")
        z = F.linear(z, sd["linear.weight"], sd["linear.bias"])
        z = _bn(z, sd, "bnembed")
        keep("embed", z)
        return z


def embed(sd: dict, depth: int, wav: torch.Tensor, norm: bool = False) -> torch.Tensor:
    """SpeakerModel.forward (sugar/models/__init__.py:79-114) for [B, L] waveforms, optional L2 norm (vectors.py:143-144)."""
    feats = logmel_fbank(wav).squeeze(1)
    e = subnet_forward(sd, depth, feats)
    return F.normalize(e, p=2, dim=1) if norm else e


def embed_ragged(sd: dict, depth: int, wavs, norm: bool = False) -> torch.Tensor:
    """Variable-length utterances the way the reference runs them: one at a time, unpadded (batch_size=1)."""
    return torch.cat([embed(sd, depth, w.unsqueeze(0), norm) for w in wavs], dim=0)


# --------------------------------------------------------------------------- scoring
def cosine_score(enroll: torch.Tensor, test: torch.Tensor) -> float:
    """calculate_score(mode='cosine') (sugar/scores.py:146-150): mean over the n x m pairwise cosines."""
    d = F.cosine_similarity(enroll.cpu().unsqueeze(-1), test.cpu().unsqueeze(-1).transpose(0, 2))
    return d.mean().item()


def norm2_score(enroll: torch.Tensor, test: torch.Tensor) -> float:
    """
    calculate_score(mode='norm2') (sugar/scores.py:141-145) -- the reference's DEFAULT mode, restated as the arithmetic
    it actually performs: F.pairwise_distance(enroll[n, D, 1], test[1, D, m]) broadcasts to [n, D, m] and takes the
    2-norm of (x1 - x2 + 1e-6) over the LAST axis, i.e. over the m test segments (not over D), giving [n, D]; the score
    is minus the mean of that.  (For m = 1 this is minus the mean absolute difference, offset by the eps.)
    """
    e = enroll.cpu().float().unsqueeze(2)                   # [n, D, 1]
    t = test.cpu().float().t().unsqueeze(0)                 # [1, D, m]
    d = (e - t + 1e-6).pow(2).sum(dim=2).sqrt()             # [n, D]
    return -d.mean().item()


def score_trials(enroll_idx, test_idx, vectors, mode="cosine") -> np.ndarray:
    """score_trials loop (sugar/scores.py:188-214) on integer utterance ids; vectors: list of [n_i, D] tensors.
    mode: "cosine", "norm2" or a list of them (then one column per mode, :151-152)."""
    fn = {"cosine": cosine_score, "norm2": norm2_score}
    if isinstance(mode, list):
        return np.array([[fn[m](vectors[int(a)], vectors[int(b)]) for m in mode] for a, b in zip(enroll_idx, test_idx)])
    return np.array([fn[mode](vectors[int(a)], vectors[int(b)]) for a, b in zip(enroll_idx, test_idx)])


def count_macs(config, T: int) -> float:
    """conv + linear MACs per utterance (SURVEY 8d formula; known answers in BASELINE.md section 2)."""
    depth, ch, ks, ccat = config
    c0 = ch[0]
    macs = 80 * ks[0] * c0 * T
    for i in range(1, depth + 1):
        ci, ki = ch[i], ks[i]
        w = ci // SCALE
        macs += 2 * c0 * (w * SCALE) * T + 7 * w * w * ki * T + 2 * c0 * (c0 // SE_REDUCTION)
    macs += depth * c0 * ccat * T + 2 * ccat * BOTTLENECK * T + 2 * ccat * 192
    return float(macs)


# --------------------------------------------------------------------------- score normalisation (8f rank 2)
def score_cohorts(cohorts: torch.Tensor, vectors: dict) -> dict:
    """score_cohorts (sugar/scores.py:77-111): per utterance, cosine of every segment vector with every cohort vector
    (F.cosine_similarity broadcast), mean over the segments -> {utterance: float32 ndarray [n_cohort]}."""
    out = {}
    for k, vec in vectors.items():
        out[k] = F.cosine_similarity(vec.unsqueeze(-1), cohorts.unsqueeze(-1).transpose(0, 2)).mean(0).numpy()
    return out


def asnorm(orig_scores, enrolls, tests, cohort: dict, ntop: int = 300, std_eps: float = 1e-8) -> np.ndarray:
    """asnorm (sugar/scores.py:332-400): top-|ntop| cohort scores per utterance (ntop < 0: the smallest) ->
    mean, sqrt(sum((v - m)^2 / |ntop|)); result 0.5 * (z-norm + t-norm) in float64."""
    mean, std = {}, {}
    for k, value in cohort.items():
        v = np.sort(np.asarray(value))
        v = v[:-ntop] if ntop < 0 else v[len(v) - ntop:]
        m = v.mean()
        mean[k] = m
        std[k] = np.sum((v - m) ** 2 / abs(ntop)) ** 0.5
    return _znorm_tnorm(orig_scores, enrolls, tests, mean, std, std_eps)


def snorm(orig_scores, enrolls, tests, cohort: dict, std_eps: float = 1e-8) -> np.ndarray:
    """snorm without nTop (sugar/scores.py:295-329): all cohort scores; the std divides by N = number of TRIALS,
    as the reference does (:295, :308)."""
    n = len(orig_scores)
    mean, std = {}, {}
    for k, value in cohort.items():
        m = value.mean()
        mean[k] = m
        std[k] = np.sum((value - m) ** 2 / n) ** 0.5
    return _znorm_tnorm(orig_scores, enrolls, tests, mean, std, std_eps)


def _znorm_tnorm(orig_scores, enrolls, tests, mean, std, std_eps):
    s = np.asarray(orig_scores, dtype=np.float64)
    zm = np.array([mean[e] for e in enrolls], dtype=np.float64)
    zs = np.array([max(std[e], std_eps) for e in enrolls], dtype=np.float64)
    tm = np.array([mean[t] for t in tests], dtype=np.float64)
    ts = np.array([max(std[t], std_eps) for t in tests], dtype=np.float64)
    return 0.5 * ((s - zm) / zs + (s - tm) / ts)
