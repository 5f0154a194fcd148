"""
NAS subnet-evaluation sweep (BASELINE config 5; the loop of tutorials/EfficientTDNN/SubnetEvaluation.ipynb cell
`ec189126` and TDNN-NAS.ipynb cell 24):

    for config in configs:
        subnet = supernet.clone(config)              # sugar/models/dynamictdnn.py:713-730
        batch_forward(train_loader, model)           # BN recalibration, sugar/models/__init__.py:201-242
        eer, dcf = veri_validate(veri_loader, model) # sugar/models/__init__.py:535-595

The reference runs the configs one after another on one GPU (and, in scripts/train_vox2_veri.py:385-403, strides
CHECKPOINTS over the GPUs: rank g evaluates models g, g + N, ... over the whole list, no collective).  Here the configs
are dealt to the ranks the same way (one process per GPU, longest first by multiply-adds, NO collective on the data
path: every rank embeds the WHOLE evaluation list with its own subnets -- extract_vectors is called un-sharded, whatever
process group is live), and on each rank the NEXT subnet is cloned and packed for the GPU on a host thread while the
current one is being evaluated (clone + pack is 8-12 ms of host work against 12-30 ms of GPU work for 4 874
utterances, tools/clone_bench.py).  The evaluation list is staged on the GPU once (PackedUtterances) and the trial
index arrays are built once, so a subnet costs its GPU time and little else.  The only exchange is one
all_gather_object of the per-config results.
"""
from concurrent.futures import ThreadPoolExecutor

import torch

from . import _lib
from .dynamictdnn import DynamicTDNN, TDNNSubnet
from .models import SpeakerModel, WrappedModel, batch_forward, veri_validate
from .transforms import LogMelFbanks
from .vectors import PackedUtterances, extract_vectors, shard_indices


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
    sub = TDNNSubnet(cfg, state, supernet.in_feats, supernet.out_embeds, precision)
    sub.to(dev)
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
                     on_result=None):
    """
    Evaluate every subnet config of `configs` (an iterable of (depth, channels, kernels, catconv)) cloned from
    `supernet` (a DynamicTDNN holding the supernet weights): optional BN recalibration over `recal_loader`
    (batch_forward), then veri_validate over `veri_loader`.

    Returns {config_key: {"eer": .., "eer_thresh": .., "mindcf": .., "mindcf_thresh": .., "rank": ..}} for ALL configs
    on every rank when `gather` (one all_gather_object), else for this rank's share.  `rank` / `world` default to
    torch.distributed's when it is initialised, else 0 / 1.  `on_result(config, result)` is called as results arrive.
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
    packed = PackedUtterances(veri_loader.dataset.utterances, dev, batch_size=max(1, int(veri_loader.batch_size)))
    veri_loader = _LoaderView(veri_loader, _TrialSnapshot(veri_loader.dataset))     # trial list parsed once
    results = {}
    pool = ThreadPoolExecutor(max_workers=1) if prefetch and len(mine) > 1 else None
    try:
        nxt = pool.submit(_prepare, supernet, mine[0], precision, dev) if pool and mine else None
        for i, cfg in enumerate(mine):
            sub = nxt.result() if nxt is not None else _prepare(supernet, cfg, precision, dev)
            nxt = pool.submit(_prepare, supernet, mine[i + 1], precision, dev) if pool and i + 1 < len(mine) else None
            model = WrappedModel(SpeakerModel(sub, transform=fb)).to(dev).eval()
            if recal_loader is not None:
                batch_forward(recal_loader, model, device=str(dev))
            with torch.no_grad():
                # un-sharded on purpose: this rank's subnet over the whole list (a live process group does not shard)
                vectors = extract_vectors(model, packed, norm=norm)
                eer, dcf = veri_validate(veri_loader, model, norm=norm, p_target=p_target, device=str(dev),
                                         vectors=vectors)
            res = {"eer": float(eer[0]), "eer_thresh": float(eer[1]), "mindcf": float(dcf[0]),
                   "mindcf_thresh": float(dcf[1]), "rank": rank}
            results[cfg] = res
            if on_result is not None:
                on_result(cfg, res)
            sub.release()                              # device weights of a finished subnet
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
    return gather_results(results, world, dev) if gather else results
