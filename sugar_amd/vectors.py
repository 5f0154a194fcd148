"""
extract_vectors with the reference's signature and return value (sugar/vectors.py:63-154):
``{file: Tensor[n_segments, 192]}`` on the device, L2-normalised when ``norm`` is True.

The reference runs whole (variable-length) utterances at batch 1 (:98-124) and equal-length segment stacks through
a DataLoader (:126-153).  Here both become packed batches: utterances are sorted by length and cut into batches
bounded by ``batch_size`` utterances and ``max_frames`` frames, each batch is ONE fused C-ABI call
(fbank + TDNN + L2 norm) with no padding, and results are scattered back to the reference's per-file layout.
"""
import torch

from . import _lib
from .transforms import num_frames


def _segments(item):
    """item['input']: [n_seg, L] (loadWAV evaluation stacks, sugar/waveforms.py:105-113) or [L] -> list of 1-D tensors."""
    x = item["input"]
    if x.dim() == 1:
        return [x]
    return [x[i] for i in range(x.shape[0])]


def plan_batches(lengths, batch_size, max_frames):
    """Sort by length, then greedy batches with <= batch_size utterances and <= max_frames frames.  Returns index lists."""
    order = sorted(range(len(lengths)), key=lambda i: lengths[i])
    batches, cur, frames = [], [], 0
    for i in order:
        t = num_frames(lengths[i])
        if cur and (len(cur) >= batch_size or frames + t > max_frames):
            batches.append(cur)
            cur, frames = [], 0
        cur.append(i)
        frames += t
    if cur:
        batches.append(cur)
    return batches


N_LANES = 2      # concurrent HIP streams of extract_vectors


class VectorTable(dict):
    """
    What extract_vectors returns: the reference's {file: Tensor[n_seg, 192]} (every value a view of one [n_rows, 192]
    matrix on the device), plus that matrix (`.matrix`), the files in order (`.files`) and the segment offsets
    (`.seg_off`, int32 [n_files + 1] on the device), so that score_trials / score_cohorts use the table as it is instead
    of concatenating thousands of views again.
    """

    def __init__(self, files, counts, matrix):
        super().__init__()
        self.files = files
        self.matrix = matrix
        seg, start = [0], 0
        for f, n in zip(files, counts):            # segments of one file are contiguous rows
            self[f] = matrix[start:start + n]
            start += n
            seg.append(start)
        self.seg_off = torch.tensor(seg, dtype=torch.int32).to(matrix.device)


class PackedUtterances:
    """
    An evaluation list staged ONCE on the GPU: utterances sorted by length, cut into batches (<= batch_size utterances,
    <= max_frames frames), each batch one flat float32 sample buffer with its index arrays.  extract_vectors accepts it
    in place of the dataset; a NAS sweep embeds the same list with every subnet and pays the host-side walk over the
    dataset (4 874 __getitem__ calls, ~45 ms) once instead of per subnet.
    """

    def __init__(self, datalst, device="cuda:0", batch_size=256, max_frames=262144):
        from .transforms import BatchPlan
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SugarError("sugar_amd runs on an MI355X only: pass device='cuda:N' (there is no CPU fallback)")
        files, counts, segs = [], [], []
        for idx in range(len(datalst)):
            item = datalst[idx]
            files.append(item["file"])
            ss = _segments(item)
            counts.append(len(ss))
            segs.extend(ss)
        self.files, self.counts, self.n_rows = files, counts, len(segs)
        lengths = [int(s.numel()) for s in segs]
        self.batches = []
        for batch in plan_batches(lengths, max(1, batch_size), max_frames):
            flat = torch.cat([segs[i].reshape(-1).to(self.device, non_blocking=True).float() for i in batch])
            self.batches.append((torch.tensor(batch, device=self.device), flat, BatchPlan([lengths[i] for i in batch],
                                                                                          self.device)))

    def __len__(self):
        return len(self.files)


def extract_vectors(model, datalst, norm=True, device="cuda:0", n_jobs=4, batch_size=1, max_frames=262144):
    """
    model   : WrappedModel / SpeakerModel from sugar_amd.models
    datalst : an indexable dataset of {"file": str, "input": Tensor[n_seg, L] or [L]} (sugar/database.py:136-142),
              or a PackedUtterances built from one (then `device`, `batch_size`, `max_frames` are the packing's)
    n_jobs  : kept for signature parity (the reference's joblib threads / DataLoader workers); unused
    Returns a VectorTable: {file: Tensor[n_seg, 192]} on the device.
    """
    packed = datalst if isinstance(datalst, PackedUtterances) else None
    device = packed.device if packed is not None else torch.device(device)
    if device.type != "cuda":
        raise _lib.SugarError("sugar_amd runs on an MI355X only: pass device='cuda:N' (there is no CPU fallback)")
    model.to(device)
    model.eval()
    spk = model.module if hasattr(model, "module") else model
    # consecutive batches alternate between two HIP streams (each with its own workspace): the HBM-bound kernels
    # of one batch run under the MFMA-bound kernels of the other (+9 % on 256 x 3 s batches) and the host-to-device
    # copies of the next batch overlap the compute of the current one
    main = torch.cuda.current_stream(device)
    lanes = [torch.cuda.Stream(device) for _ in range(N_LANES)]
    for lane in lanes:
        lane.wait_stream(main)
    if packed is not None:
        files, counts = packed.files, packed.counts
        out = torch.empty(packed.n_rows, 192, dtype=torch.float32, device=device)
        sub = spk._subnet()
        for k, (idx, flat, plan) in enumerate(packed.batches):
            with torch.cuda.stream(lanes[k % N_LANES]):
                out[idx] = spk._embed_flat(sub, flat, plan, bool(norm))
    else:
        files, counts, segs = [], [], []
        for idx in range(len(datalst)):
            item = datalst[idx]
            files.append(item["file"])
            ss = _segments(item)
            counts.append(len(ss))
            segs.extend(ss)
        lengths = [int(s.numel()) for s in segs]
        out = torch.empty(len(segs), 192, dtype=torch.float32, device=device)
        for k, batch in enumerate(plan_batches(lengths, max(1, batch_size), max_frames)):
            with torch.cuda.stream(lanes[k % N_LANES]):
                wavs = [segs[i].to(device, non_blocking=True) for i in batch]
                emb = spk.embed_ragged(wavs, normalize=bool(norm))
                out[torch.tensor(batch, device=device)] = emb
    for lane in lanes:
        main.wait_stream(lane)
    return VectorTable(files, counts, out)


def gather_vectors(local, world_size=None):
    """
    Multi-GPU exchange step: every rank embedded a shard of the utterances; one all-gather of the
    [n_local, 192] float32 blocks (RCCL over xGMI on GPUs, gloo in CPU tests) gives every rank all vectors.
    ``local``: Tensor [n_local, D]; shards may differ in size (padded to the max for the collective).
    Returns the list of per-rank tensors in rank order.
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    world = dist.get_world_size() if world_size is None else world_size
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = local if local.shape[0] == m else torch.cat([local, local.new_zeros(m - local.shape[0], local.shape[1])])
    buf = local.new_empty(world * m, local.shape[1])
    dist.all_gather_into_tensor(buf, pad.contiguous())
    return [buf[r * m:r * m + sizes[r]] for r in range(world)]


def shard_indices(n_items, rank, world, weights=None):
    """Deal items to ranks: longest-first round-robin when weights (lengths) are given, else strided."""
    if weights is None:
        return list(range(rank, n_items, world))
    order = sorted(range(n_items), key=lambda i: -weights[i])
    return sorted(order[rank::world])
