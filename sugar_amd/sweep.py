"""
NAS subnet-evaluation sweep (BASELINE config 5; the loop of tutorials/EfficientTDNN/SubnetEvaluation.ipynb cell
`ec189126` and TDNN-NAS.ipynb cell 24):

    for config in configs:
        subnet = supernet.clone(config)              # sugar/models/dynamictdnn.py:713-730
        batch_forward(train_loader, model)           # BN recalibration, sugar/models/__init__.py:201-242
        eer, dcf = veri_validate(veri_loader, model) # sugar/models/__init__.py:535-595 (here: its two halves, see below)

The reference runs the configs one after another on one GPU (and, in scripts/train_vox2_veri.py:385-403, strides
CHECKPOINTS over the GPUs: rank g evaluates models g, g + N, ... over the whole list, no collective).  Here the configs
are dealt to the ranks the same way (one process per GPU, longest first by multiply-adds, NO collective on the data
path: every rank embeds the WHOLE evaluation list with its own subnets -- extract_vectors is called un-sharded, whatever
process group is live), and on each rank the NEXT subnets are cloned and packed for the GPU on host threads while the
current one is being evaluated (clone + pack is 8-10 ms of host work against ~12 ms of GPU work for 4 874 utterances),
the extraction + scoring of subnet i are enqueued before the host computes EER / minDCF of subnet i - 1, and nothing
on the way waits for the device (packed weights go up in one asynchronous copy, csrc/arena.hip; the vector table's
index is built once).  The evaluation list is staged on the GPU once (PackedUtterances) and the trial
index arrays are built once, so a subnet costs its GPU time and little else.  The only exchange is one
all_gather_object of the per-config results.
"""
from concurrent.futures import ThreadPoolExecutor
from time import perf_counter as clock

import torch

from . import _lib
from .dynamictdnn import DynamicTDNN, TDNNSubnet
from .metrics import calculate_eer, calculate_mindcf
from .models import SpeakerModel, WrappedModel, batch_forward
from .scores import score_trials_device
from .transforms import LogMelFbanks
from .vectors import AUTO_BATCH, PackedUtterances, extract_vectors, shard_indices


def config_key(config):
    """Hashable form of a subnet config (depth, channels, kernels, catconv)."""
    depth, ch, ks, ccat = config
    return (int(depth), tuple(int(c) for c in ch), tuple(int(k) for k in ks), int(ccat))


def _cost(supernet, config, frames=300):
    """Multiply-adds of `config` by the reference's own count (dynamictdnn.py:735-765); restores the active config."""
    saved = supernet.config
    try:
        supernet.config = config
        return supernet.count_ops((1, supernet.in_feats, frames))
    finally:
        supernet.config = saved


def _prepare(supernet, config, precision, dev):
    """clone(config) without touching the supernet's active config (safe on a host thread), packed for `dev`."""
    cfg, state = supernet.materialize(config)
    sub = TDNNSubnet(cfg, state, supernet.in_feats, supernet.out_embeds, precision, adopt=True)
    sub.to(dev)
    sub.bn_lazy = True                                 # evaluated and thrown away: nobody reads its buffers after recalibration
    sub.handle(dev)                                    # sugar_subnet_create: fold BN, pack, upload
    return sub


def shard_configs(supernet, configs, rank, world):
    """Indices of `configs` this rank evaluates: longest-first round-robin by multiply-adds (same rule as utterances)."""
    return shard_indices(len(configs), rank, world, weights=[_cost(supernet, c) for c in configs])


def gather_results(local, world=None, device=None):
    """{config_key: result} of this rank -> the union over all ranks (one all_gather_object; identity without a
    process group).  `device`: this rank's GPU -- under backend "nccl" (RCCL) the object gather serialises through
    torch.cuda.current_device(), which must be the rank's own card."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()):
        return dict(local)
    world = dist.get_world_size() if world is None else world
    if world <= 1:
        return dict(local)
    if world != dist.get_world_size():
        raise _lib.SugarError("gather_results(world=%d) under a process group of %d ranks" % (world, dist.get_world_size()))
    parts = [None] * world
    dev = torch.device(device) if device is not None else None
    if dev is not None and dev.type == "cuda":
        with torch.cuda.device(dev):
            dist.all_gather_object(parts, local)
    else:
        dist.all_gather_object(parts, local)
    out = {}
    for p in parts:
        out.update(p)
    return out


class _TrialSnapshot:
    """The sweep's own view of a trial list: labels / enrolls / tests read ONCE, so that score_trials' index memo (which
    only trusts VerificationTrials-shaped objects) holds for every subnet whatever container the caller passed."""

    def __init__(self, dataset):
        self.utterances = dataset.utterances
        if all(isinstance(getattr(dataset, a, None), list) for a in ("labels", "enrolls", "tests")):
            self.labels, self.enrolls, self.tests = list(dataset.labels), list(dataset.enrolls), list(dataset.tests)
        else:
            rows = [(tr["label"], tr["enroll"], tr["test"]) for tr in dataset]
            self.labels = [r[0] for r in rows]
            self.enrolls = [r[1] for r in rows]
            self.tests = [r[2] for r in rows]

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, i):
        return {"label": self.labels[i], "enroll": self.enrolls[i], "test": self.tests[i]}

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class _LoaderView:
    def __init__(self, loader, dataset):
        self.dataset, self.batch_size = dataset, loader.batch_size
        self.num_workers = getattr(loader, "num_workers", 0)


def evaluate_subnets(supernet, configs, veri_loader, recal_loader=None, transform=None, device="cuda:0",
                     precision="bf16", norm=True, p_target=0.05, prefetch=True, rank=None, world=None, gather=True,
                     on_result=None, timings=None):
    """
    Evaluate every subnet config of `configs` (an iterable of (depth, channels, kernels, catconv)) cloned from
    `supernet` (a DynamicTDNN holding the supernet weights): optional BN recalibration over `recal_loader`
    (batch_forward), then veri_validate over `veri_loader`.

    Returns {config_key: {"eer": .., "eer_thresh": .., "mindcf": .., "mindcf_thresh": .., "rank": ..}} for ALL configs
    on every rank when `gather` (one all_gather_object), else for this rank's share.  `rank` / `world` default to
    torch.distributed's when it is initialised, else 0 / 1.  `on_result(config, result)` is called as results arrive
    (one subnet behind the GPU: the metrics of subnet i - 1 are computed while subnet i runs).
    `prefetch`: subnets cloned and packed ahead on host threads (True: 2, an int: that many, False / 0: none).
    `timings`: a dict that receives the launching thread's seconds per phase (diagnostics, tools/sweep_bench.py).
    """
    if not isinstance(supernet, DynamicTDNN):
        raise TypeError("evaluate_subnets expects the DynamicTDNN supernet, got %s" % type(supernet).__name__)
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.SugarError("sugar_amd runs on an MI355X only: pass device='cuda:N' (there is no CPU fallback)")
    dist = torch.distributed
    live = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank() if live else 0
    if world is None:
        world = dist.get_world_size() if live else 1
    configs = [config_key(c) for c in configs]
    mine = [configs[i] for i in shard_configs(supernet, configs, rank, world)]
    fb = transform if transform is not None else LogMelFbanks()
    # the evaluation list is walked and staged on the GPU once for all subnets
    bs = int(veri_loader.batch_size)                   # (1 = the reference's whole-utterance mode: the packer chooses)
    packed = PackedUtterances(veri_loader.dataset.utterances, dev, batch_size=bs if bs > 1 else AUTO_BATCH,
                              cache_features=transform is None or isinstance(transform, LogMelFbanks))
    veri_loader = _LoaderView(veri_loader, _TrialSnapshot(veri_loader.dataset))     # trial list parsed once
    results = {}
    # subnets cloned + packed ahead on host threads: True = 2 (one ahead left the launching thread waiting 6 ms per subnet
    # for the next one -- both sides run Python; two ahead: 17.5-19.1 -> 16.1-16.3 ms per subnet), an int = that many
    depth_q = 2 if prefetch is True else max(0, int(prefetch))
    pool = ThreadPoolExecutor(max_workers=depth_q) if depth_q and len(mine) > 1 else None

    retired = []                                       # finished subnets whose packed weights are still on the device

    def finish(cfg, sub, host_scores, copied, labels):
        """Host half of veri_validate for a subnet whose scores are on their way: wait for ITS copy (an event, not the
        stream: the next subnet's work is already queued behind it), EER and minDCF (models.veri_validate, same calls)."""
        copied.synchronize()
        scs = host_scores.double().numpy()
        eer = calculate_eer(labels, scs)
        dcf = calculate_mindcf(labels, scs, p_target=p_target)
        res = {"eer": float(eer[0]), "eer_thresh": float(eer[1]), "mindcf": float(dcf[0]),
               "mindcf_thresh": float(dcf[1]), "rank": rank}
        results[cfg] = res
        if on_result is not None:
            on_result(cfg, res)
        # its kernels have run: the workspaces go back to torch's allocator now; the packed weights are freed in
        # batches -- hipFree waits for the whole device, i.e. for the NEXT subnet's extraction, once per call
        sub.drop_workspaces()
        retired.append(sub)
        if len(retired) >= 32:
            while retired:
                retired.pop().release()

    # Software pipeline over the subnets: the extraction + scoring of subnet i are ENQUEUED, then the host computes the
    # metrics of subnet i - 1 while the GPU works (scoring + EER / minDCF are ~5 ms of host time per subnet during which
    # the GPU used to idle: 12 ms of GPU work per subnet took 22 ms of wall-clock), and the host thread clones and
    # packs subnet i + 1.
    pending = None
    try:
        ahead = []                                     # futures of the next subnets, in order
        if pool:
            for c in mine[:depth_q]:
                ahead.append(pool.submit(_prepare, supernet, c, precision, dev))
        for i, cfg in enumerate(mine):
            t0 = clock()
            sub = ahead.pop(0).result() if pool else _prepare(supernet, cfg, precision, dev)
            if pool and i + depth_q < len(mine):
                ahead.append(pool.submit(_prepare, supernet, mine[i + depth_q], precision, dev))
            t1 = clock()
            model = WrappedModel(SpeakerModel(sub, transform=fb)).to(dev).eval()
            if recal_loader is not None:
                batch_forward(recal_loader, model, device=str(dev))
            t2 = clock()
            with torch.no_grad():
                # un-sharded on purpose: this rank's subnet over the whole list (a live process group does not shard)
                vectors = extract_vectors(model, packed, norm=norm)
                t3 = clock()
                scores_dev, labels, _ = score_trials_device(veri_loader.dataset, vectors, mode="cosine")
                host_scores = torch.empty(scores_dev.shape, dtype=scores_dev.dtype, pin_memory=True)
                host_scores.copy_(scores_dev, non_blocking=True)
                copied = torch.cuda.Event()
                copied.record(torch.cuda.current_stream(dev))
            t4 = clock()
            if pending is not None:
                finish(*pending)
            t5 = clock()
            pending = (cfg, sub, host_scores, copied, labels)
            if timings is not None:
                for k, v in (("wait_subnet", t1 - t0), ("wrap_recal", t2 - t1), ("launch_extract", t3 - t2),
                             ("launch_score", t4 - t3), ("finish_previous", t5 - t4)):
                    timings[k] = timings.get(k, 0.0) + v
        if pending is not None:
            done, pending = pending, None
            finish(*done)
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
            for f in ahead:                            # (an exception on the way: subnets packed ahead and never used)
                if not f.cancelled() and f.exception() is None:
                    retired.append(f.result())
        if pending is not None or retired:             # (an exception on the way: the subnet in flight as well)
            torch.cuda.synchronize(dev)
            for sub in retired + ([pending[1]] if pending is not None else []):
                sub.release()
    return gather_results(results, world, dev) if gather else results
