"""
Predictor-guided subnet search (sugar/models/dynamictdnn.py:951-1568 and TDNN-NAS.ipynb cells 16-19):

    archcodec = ArchCoder(len(cspace), dconfig, config2d)          # subnet config <-> feature vector
    acc_pred  = AccuracyPredictor(ndims, pretrained="EER.48.Coarse.pth.tar")
    problem   = GPULatencyConstraint(archcodec, acc_pred, TDNNLatency(table), space_mode="Coarse", threshold=15)
    subnet, info = ea_solve("EER", problem, MAXGEN=200)

Host-side search logic, no GPU kernel: the GPU enters through the latency table (`TDNNLatency.measure`, latency.py)
and the subnet evaluations that train the predictor (`evaluate_subnets`, sweep.py).  numpy only -- torch is used to
read / write the reference's predictor checkpoint format ({'model': ..., 'ynorm': ...}) and nothing else.

Pinned to the reference's own classes by tests/golden/nas.npz (oracle/make_golden_nas.py): ArchCoder.encode in its
three modes, config2d, AccuracyPredictor.predict, MinimizeError.encode.  NOT pinned (said so in the tests):
  * ArchCoder.decode -- the reference's uses `np.int` and raises on numpy >= 1.24; restated from its intent, round trips;
  * AccuracyPredictor.fit -- a stochastic training loop (shuffled mini-batches); same objective, optimiser and
    schedule, checked by what it learns;
  * ea_solve -- the reference delegates to geatpy's `soea_SEGA_templet`, which is not installed: the loop here is an
    own strengthened-elitist GA over the same integer genome and constraint handling, checked against exhaustive search.
"""
import os

import numpy as np


# ------------------------------------------------------------------------------------------------------------ ArchCoder
def config2d(config):
    """Flat config list [depth, widths (depth+1), kernels (depth+1), cat] -> dict of the 12 search dimensions
    (dynamictdnn.py:1070-1087)."""
    depth = config[0]
    if len(config) != 1 + 2 + depth * 2 + 1:
        raise AssertionError("config %r does not have 1 + 2 (depth + 1) + 1 entries" % (config,))
    d = {"depth": depth}
    for i in range(5):
        d["width%d" % i] = None
    for i in range(5):
        d["kernel%d" % i] = None
    d["cat"] = None
    for i, w in enumerate(config[1: depth + 2]):
        d["width%d" % i] = w
    for i, k in enumerate(config[depth + 2: 2 * depth + 3]):
        d["kernel%d" % i] = k
    d["cat"] = config[-1]
    return d


class ArchCoder(object):
    """
    Encode a subnet configuration as a feature vector and decode it (dynamictdnn.py:951-1067).
    mode None: one-hot per dimension; 'b': binary index (1-based, 0 = absent); 'c': (index + 1) / n_choices.
    """

    def __init__(self, nDim, dconfig, config2d=None, mode=None):
        if mode not in (None, "b", "c"):
            raise AssertionError("mode must be None, 'b' or 'c'")
        if nDim != len(dconfig):
            raise AssertionError("nDim %d != %d dimensions" % (nDim, len(dconfig)))
        self.config2d = config2d if config2d is not None else (lambda x: x)
        self.nDim, self.mode = nDim, mode
        self.arch, self.darch, self.archd, self.ndims = {}, {}, {}, []
        self.nfeat = 0
        for key, value in dconfig.items():
            value = sorted(set(value))
            self.arch[key] = value
            self.ndims.append(key)
            self.nfeat += self._width(len(value))
            self.darch[key] = {v: i for i, v in enumerate(value)}
            self.archd[key] = {i: v for i, v in enumerate(value)}

    def _width(self, n):
        return n if self.mode is None else (len(format(n, "b")) if self.mode == "b" else 1)

    def encode(self, config):
        dconfig = self.config2d(config)
        if self.nDim != len(dconfig):
            raise AssertionError("config has %d dimensions, coder %d" % (len(dconfig), self.nDim))
        feat = np.zeros(self.nfeat, dtype=np.float32)
        start = 0
        for key in self.ndims:
            n = len(self.darch[key])
            v = dconfig[key]
            if v is not None:
                if v not in self.darch[key]:
                    raise AssertionError("v: {} not in darch[{}]".format(v, key))
                i = self.darch[key][v]
                if self.mode is None:
                    feat[start + i] = 1.0
                elif self.mode == "b":
                    for j, bit in enumerate(format(i + 1, "b").zfill(self._width(n))):
                        feat[start + j] = int(bit)
                else:
                    feat[start] = (i + 1) * 1.0 / n
            start += self._width(n)
        return feat

    def decode(self, feat, typ=0):
        """typ 0: dict of the dimensions (None = absent); typ 1: flat list of the present values."""
        feat = np.asarray(feat)
        if self.nfeat != feat.shape[0] or typ not in (0, 1):
            raise AssertionError("bad feature length / typ")
        if self.mode != "c":
            feat = feat.astype(np.int64)
        dconfig, start = {}, 0
        for key in self.ndims:
            n = len(self.darch[key])
            part = feat[start: start + self._width(n)]
            if self.mode is None:
                idx = np.where(part == 1)[0]
                v = self.archd[key][int(idx[0])] if len(idx) == 1 else None
            elif self.mode == "b":
                i = int("".join(str(int(b)) for b in part), 2)
                v = None if i == 0 else self.archd[key][i - 1]
            else:
                x = float(part[0])
                v = None if x == 0 else self.archd[key][int(round(x * n)) - 1]
            dconfig[key] = v
            start += self._width(n)
        if typ == 0:
            return dconfig
        return [dconfig[key] for key in self.ndims if dconfig[key] is not None]


# ---------------------------------------------------------------------------------------------------- AccuracyPredictor
class AccuracyPredictor(object):
    """
    BatchNorm1d(in) -> Linear(in, 400) -> ReLU -> Linear(400, 400) -> ReLU -> Linear(400, 400) -> ReLU -> Linear(400, 1)
    predicting a BatchNorm-normalised target (`ynorm`), un-normalised by predict() (dynamictdnn.py:1090-1172).
    Parameters live in numpy under the reference's state-dict keys ('0.weight' ... '7.bias' / ynorm 'weight' ...);
    `pretrained` reads the reference's checkpoint ({'model': ..., 'ynorm': ...}, torch.load with weights_only=True),
    `save` writes it.  As in the reference, `transform` is accepted and has no effect on the forward (its result is
    discarded there, dynamictdnn.py:1135-1136).
    """
    HIDDEN = 400
    EPS = 1e-5
    MOMENTUM = 0.1

    def __init__(self, in_feats, pretrained=None, device="cpu", transform=False, verbose=False, seed=0):
        self.in_feats, self.device, self.transform, self.verbose = in_feats, device, transform, verbose
        rng = np.random.default_rng(seed)
        self.model = {"0.weight": np.ones(in_feats, np.float32), "0.bias": np.zeros(in_feats, np.float32),
                      "0.running_mean": np.zeros(in_feats, np.float32), "0.running_var": np.ones(in_feats, np.float32),
                      "0.num_batches_tracked": np.array(0, np.int64)}
        for idx, (fi, fo) in zip((1, 3, 5, 7), ((in_feats, 400), (400, 400), (400, 400), (400, 1))):
            bound = 1.0 / np.sqrt(fi)                      # nn.Linear's default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            self.model["%d.weight" % idx] = rng.uniform(-bound, bound, (fo, fi)).astype(np.float32)
            self.model["%d.bias" % idx] = rng.uniform(-bound, bound, fo).astype(np.float32)
        self.ynorm = {"weight": np.ones(1, np.float32), "bias": np.zeros(1, np.float32),
                      "running_mean": np.zeros(1, np.float32), "running_var": np.ones(1, np.float32),
                      "num_batches_tracked": np.array(0, np.int64)}
        self.fit_process = {}
        if pretrained:
            if not os.path.exists(pretrained):
                raise FileNotFoundError(pretrained)
            import torch
            sd = torch.load(pretrained, map_location="cpu", weights_only=True)
            self.load_state(sd["model"], sd["ynorm"])
        elif verbose:
            print("MUST train the prodictor FIRST!")

    def load_state(self, model, ynorm):
        for dst, src in ((self.model, model), (self.ynorm, ynorm)):
            for k in dst:
                v = src[k]
                v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
                if v.shape != dst[k].shape:
                    raise ValueError("predictor parameter %s has shape %s, expected %s" % (k, v.shape, dst[k].shape))
                dst[k] = v.astype(dst[k].dtype)

    # ---- forward (eval: running statistics) --------------------------------------------------------------------------
    def _forward_eval(self, x):
        m = self.model
        h = (x - m["0.running_mean"]) / np.sqrt(m["0.running_var"] + self.EPS) * m["0.weight"] + m["0.bias"]
        for idx in (1, 3, 5):
            h = np.maximum(h @ m["%d.weight" % idx].T + m["%d.bias" % idx], 0.0)
        return h @ m["7.weight"].T + m["7.bias"]

    def predict(self, inputs):
        """[n, in_feats] (or [in_feats]) features -> predicted metric, 1-D ndarray (0-d for one row, as the reference)."""
        x = np.asarray(inputs.detach().cpu().numpy() if hasattr(inputs, "detach") else inputs, dtype=np.float32)
        if x.shape[-1] != self.in_feats:
            raise AssertionError("inputs have %d features, predictor %d" % (x.shape[-1], self.in_feats))
        y = self._forward_eval(x.reshape(-1, self.in_feats))
        bn = self.ynorm                                     # invert the target's BatchNorm (dynamictdnn.py:1168-1169)
        y = (y - bn["bias"]) / (bn["weight"] + self.EPS) * np.sqrt(bn["running_var"] + self.EPS) + bn["running_mean"]
        return np.squeeze(y.astype(np.float32))

    # ---- training ----------------------------------------------------------------------------------------------------
    def fit(self, X, y, n_epoch=50, lr=1e-3, ce="mse", optimizer="adam", scheduler="steplr", alpha=1e-3, gamma=0.5,
            cycle=2, seed=0):
        """Mini-batch training with the reference's recipe (dynamictdnn.py:1174-1264): batches of 256, shuffled, last
        partial batch dropped; loss between the output and the batch-normalised target; Adam (or SGD momentum 0.9) with
        weight decay `alpha`; StepLR(n_epoch // cycle, gamma).  Returns the final mean absolute error on (X, y)."""
        if ce not in ("mse", "l1", "smoothl1") or optimizer not in ("sgd", "adam") or scheduler != "steplr":
            raise AssertionError("ce in mse|l1|smoothl1, optimizer in sgd|adam, scheduler steplr")
        X = np.asarray(X, np.float32)
        y = np.asarray(y, np.float32).reshape(-1, 1)
        if X.shape[-1] != self.in_feats or X.shape[0] < 256:
            raise AssertionError("need >= 256 samples of %d features" % self.in_feats)
        rng = np.random.default_rng(seed)
        names = [k for k in self.model if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
        st = self.fit_process.setdefault("state", {k: [np.zeros_like(self.model[k]), np.zeros_like(self.model[k])]
                                                   for k in names})
        step = self.fit_process.get("step", 0)
        cur_lr = self.fit_process.get("lr", lr)
        m = self.model
        for epoch in range(n_epoch):
            order = rng.permutation(X.shape[0])
            for b0 in range(0, X.shape[0] - 255, 256):
                xb, yb = X[order[b0:b0 + 256]], y[order[b0:b0 + 256]]
                n = 256.0
                # forward, train mode
                mu, var = xb.mean(0), xb.var(0)
                inv = 1.0 / np.sqrt(var + self.EPS)
                xh = (xb - mu) * inv
                a0 = xh * m["0.weight"] + m["0.bias"]
                m["0.running_mean"] = (1 - self.MOMENTUM) * m["0.running_mean"] + self.MOMENTUM * mu
                m["0.running_var"] = (1 - self.MOMENTUM) * m["0.running_var"] + self.MOMENTUM * var * n / (n - 1)
                acts, h = [a0], a0
                for idx in (1, 3, 5):
                    h = np.maximum(h @ m["%d.weight" % idx].T + m["%d.bias" % idx], 0.0)
                    acts.append(h)
                out = h @ m["7.weight"].T + m["7.bias"]
                ymu, yvar = yb.mean(0), yb.var(0)
                yn = (yb - ymu) / np.sqrt(yvar + self.EPS) * self.ynorm["weight"] + self.ynorm["bias"]
                self.ynorm["running_mean"] = ((1 - self.MOMENTUM) * self.ynorm["running_mean"] + self.MOMENTUM * ymu).astype(np.float32)
                self.ynorm["running_var"] = ((1 - self.MOMENTUM) * self.ynorm["running_var"] + self.MOMENTUM * yvar * n / (n - 1)).astype(np.float32)
                d = out - yn
                if ce == "mse":
                    g = 2.0 * d / n
                elif ce == "l1":
                    g = np.sign(d) / n
                else:
                    g = np.where(np.abs(d) < 1.0, d, np.sign(d)) / n
                # backward
                grads = {}
                for li, idx in zip((3, 2, 1, 0), (7, 5, 3, 1)):
                    grads["%d.weight" % idx] = g.T @ acts[li]
                    grads["%d.bias" % idx] = g.sum(0)
                    g = g @ m["%d.weight" % idx]
                    if li > 0:
                        g = g * (acts[li] > 0)
                grads["0.weight"] = (g * xh).sum(0)
                grads["0.bias"] = g.sum(0)
                step += 1
                for k in names:
                    gk = grads[k].astype(np.float32) + alpha * m[k]
                    if optimizer == "adam":
                        st[k][0] = 0.9 * st[k][0] + 0.1 * gk
                        st[k][1] = 0.999 * st[k][1] + 0.001 * gk * gk
                        mh = st[k][0] / (1 - 0.9 ** step)
                        vh = st[k][1] / (1 - 0.999 ** step)
                        m[k] = (m[k] - cur_lr * mh / (np.sqrt(vh) + 1e-8)).astype(np.float32)
                    else:
                        st[k][0] = 0.9 * st[k][0] + gk
                        m[k] = (m[k] - cur_lr * st[k][0]).astype(np.float32)
            if epoch < n_epoch - 1 and (epoch + 1) % max(1, n_epoch // cycle) == 0:
                cur_lr *= gamma
        self.fit_process.update(step=step, lr=cur_lr)
        l1 = float(np.abs(self.predict(X) - y[:, 0]).mean())
        if self.verbose:
            print("Train Epoch %d at L1 Error %.3f and LR %.6f" % (n_epoch, l1, cur_lr))
        return l1

    def save(self, model_path):
        if os.path.exists(model_path):
            raise FileExistsError(model_path)
        import torch
        torch.save({"model": {k: torch.from_numpy(np.array(v)) for k, v in self.model.items()},
                    "ynorm": {k: torch.from_numpy(np.array(v)) for k, v in self.ynorm.items()}}, model_path)


# ------------------------------------------------------------------------------------------------- the search problem
class MinimizeError(object):
    """
    Minimise the predicted error rate over the 12-gene integer genome of tdnn8m2g under a constraint
    (dynamictdnn.py:1444-1529).  Genes: depth, 5 width indices, 5 kernel indices, catconv index; `encode` maps genomes to
    flat config lists [depth, widths (depth+1), kernels (depth+1), cat] truncated to the genome's depth.
    Subclasses implement constraints(subnets) -> indices of the subnets that violate it.
    """
    CHANNELS = {0: 128, 1: 176, 2: 256, 3: 384, 4: 512}
    CATCONV = {0: 384, 1: 536, 2: 768, 3: 1152, 4: 1536}

    def __init__(self, arch_codec, acc_pred, lat_table, space_mode="Coarse", threshold=50, name="Test"):
        if space_mode not in ("Coarse", "Fine"):
            raise AssertionError("space_mode must be 'Coarse' or 'Fine'")
        self.name, self.M, self.maxormins, self.Dim = name, 1, [1], 12
        self.varTypes = [1] * 12
        self.lb = [2, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 0]
        self.ub = [4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 4] if space_mode == "Coarse" else [4, 48, 48, 48, 48, 48, 3, 3, 3, 3, 3, 144]
        self.space_mode, self.arch_codec, self.acc_pred, self.lat_table = space_mode, arch_codec, acc_pred, lat_table
        self.threshold = threshold

    def evaluate(self, genomes):
        """(objective [n], violated [n] bool) of integer genomes [n, 12] (the reference's aimFunc)."""
        subnets = self.encode(np.array(genomes, dtype=np.int64))
        obj = np.atleast_1d(self.acc_pred.predict(np.array([self.arch_codec.encode(x) for x in subnets])))
        violated = np.zeros(len(subnets), dtype=bool)
        violated[self.constraints(subnets)] = True
        return obj.astype(np.float64), violated

    def constraints(self, subnets):
        raise NotImplementedError

    def encode(self, X):
        X = np.array(X, dtype=np.int64)
        single = X.ndim == 1
        if single:
            X = X[np.newaxis, :]
        X = X.copy()
        if self.space_mode == "Coarse":
            X[:, 1:6] = np.vectorize(self.CHANNELS.get)(X[:, 1:6])
            X[:, 6:11] = X[:, 6:11] * 2 - 1
            X[:, 11] = np.vectorize(self.CATCONV.get)(X[:, 11])
        else:
            X[:, 1:6] = X[:, 1:6] * 8 + 128
            X[:, 6:11] = X[:, 6:11] * 2 - 1
            X[:, 11] = X[:, 11] * 8 + 384
        out = [[x[0]] + x[1: 2 + x[0]] + x[6: 7 + x[0]] + [x[-1]] for x in X.tolist()]
        return out[0] if single else out


def split_config(flat):
    """Flat config list -> (depth, widths, kernels, catconv), the form DynamicTDNN.clone and TDNNLatency.query take."""
    d = flat[0]
    return (d, flat[1: 2 + d], flat[2 + d: 3 + 2 * d], flat[-1])


class GPULatencyConstraint(MinimizeError):
    """The constraint of TDNN-NAS.ipynb cell 19: table latency of the subnet below `threshold` (ms)."""

    def __init__(self, arch_codec, acc_pred, lat_table, space_mode="Coarse", threshold=50, name="GPULatency"):
        MinimizeError.__init__(self, arch_codec, acc_pred, lat_table, space_mode, threshold, name)

    def constraints(self, subnets):
        lat = np.array([self.lat_table.query(split_config(x)) for x in subnets])
        return np.where(lat >= self.threshold)


def ea_solve(target, problem, encode="BG", NIND=50, MAXGEN=800, Pm=0.1, verbose=True, seed=0, Pc=0.7, patience=None):
    """
    Strengthened-elitist genetic search over the problem's integer genome (the role of geatpy's soea_SEGA_templet in
    dynamictdnn.py:1531-1568): tournament parents, uniform crossover (probability Pc), per-gene resampling mutation
    (probability Pm), parents and offspring merged (one copy per genome) and the best NIND kept -- every feasible
    individual ranks before every infeasible one, then by the objective.  `encode` is accepted for
    signature parity (the genome is integer-coded here).  Returns (subnet, {'ACC': objective, 'Latency': ms}) with
    subnet = (depth, widths, kernels, cat), or None when no feasible individual was found.
    """
    rng = np.random.default_rng(seed)
    lb, ub = np.array(problem.lb), np.array(problem.ub)
    pop = rng.integers(lb, ub + 1, size=(NIND, problem.Dim))
    obj, bad = problem.evaluate(pop)
    n_eval = NIND
    key = lambda o, b: np.lexsort((o, b))                     # feasible (b False) first, then ascending objective
    best_hist, stall = None, 0
    for gen in range(MAXGEN):
        # binary tournaments on (feasibility, objective)
        a, b = rng.integers(0, NIND, NIND), rng.integers(0, NIND, NIND)
        a_wins = (bad[a] < bad[b]) | ((bad[a] == bad[b]) & (obj[a] <= obj[b]))
        parents = pop[np.where(a_wins, a, b)]
        mates = parents[rng.permutation(NIND)]
        cross = (rng.random((NIND, 1)) < Pc) & (rng.random((NIND, problem.Dim)) < 0.5)
        kids = np.where(cross, mates, parents)
        mut = rng.random((NIND, problem.Dim)) < Pm
        kids = np.where(mut, rng.integers(lb, ub + 1, size=(NIND, problem.Dim)), kids)
        kobj, kbad = problem.evaluate(kids)
        n_eval += NIND
        allp, allo, allb = np.concatenate([pop, kids]), np.concatenate([obj, kobj]), np.concatenate([bad, kbad])
        # one copy of each genome, so the elite cannot fill the population with clones
        _, first = np.unique(allp, axis=0, return_index=True)
        first.sort()
        allp, allo, allb = allp[first], allo[first], allb[first]
        keep = key(allo, allb)[:NIND]
        pop, obj, bad = allp[keep], allo[keep], allb[keep]
        if len(pop) < NIND:                                    # fewer distinct genomes than slots: refill at random
            extra = rng.integers(lb, ub + 1, size=(NIND - len(pop), problem.Dim))
            eo, eb = problem.evaluate(extra)
            n_eval += len(extra)
            pop, obj, bad = np.concatenate([pop, extra]), np.concatenate([obj, eo]), np.concatenate([bad, eb])
        if verbose and gen % 50 == 0:
            print("gen %4d  best %s  feasible %d / %d" % (gen, obj[0] if not bad[0] else None, int((~bad).sum()), NIND))
        cur = None if bad[0] else float(obj[0])
        stall = stall + 1 if cur == best_hist else 0
        best_hist = cur
        if patience is not None and stall >= patience:
            break
    if verbose:
        print("The number of evaluation: %s" % n_eval)
    order = key(obj, bad)
    if bad[order[0]]:
        if verbose:
            print("No solutions!")
        return None
    flat = problem.encode(pop[order[0]])
    subnet = split_config(flat)
    lat = problem.lat_table.query(subnet) if problem.lat_table else -1.0
    if verbose:
        print("The Optimal %s: %.2f%%" % (target, obj[order[0]]))
        print("The optimal solution: {}: {:.1f}ms".format(subnet, lat))
    return subnet, {"ACC": float(obj[order[0]]), "Latency": lat}
