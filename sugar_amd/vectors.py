"""
extract_vectors with the reference's signature and return value (sugar/vectors.py:63-154):
``{file: Tensor[n_segments, 192]}`` on the device, L2-normalised when ``norm`` is True.

The reference runs whole (variable-length) utterances at batch 1 (:98-124) and equal-length segment stacks through
a DataLoader (:126-153).  Here both become packed batches: utterances are sorted by length and cut into batches
bounded by ``batch_size`` utterances and ``max_frames`` frames, each batch is ONE fused C-ABI call
(fbank + TDNN + L2 norm) with no padding, and results are scattered back to the reference's per-file layout.
"""
import os

import numpy as np
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


AUTO_BATCH = 256  # utterances per packed call when the caller passes the reference's batch_size = 1 (see extract_vectors)
N_LANES = 2      # concurrent HIP streams of extract_vectors
_LANES = {}      # device index -> the streams, created once (a subnet keeps one workspace per (device, stream))


def stream_lanes(device, n=None):
    """The first ``n`` (default N_LANES) of this process's extraction streams on ``device``, created on first use."""
    n = N_LANES if n is None else n
    idx = device.index if device.index is not None else torch.cuda.current_device()
    lanes = _LANES.setdefault(idx, [])
    while len(lanes) < n:
        lanes.append(torch.cuda.Stream(device))
    return lanes[:n]


class VectorTable(dict):
    """
    What extract_vectors returns: the reference's {file: Tensor[n_seg, 192]} (every value a view of one [n_rows, 192]
    matrix on the device), plus that matrix (`.matrix`), the files in order (`.files`) and the segment offsets
    (`.seg_off`, int32 [n_files + 1] on the device), so that score_trials / score_cohorts use the table as it is instead
    of concatenating thousands of views again.

    The views are made on first use: building 4 874 tensor views up front cost 22 ms of host time per call -- more than
    the GPU needs to embed the list with a small subnet, and the NAS sweep (which scores through `.matrix`) was bound by
    it.  `table[file]`, `in`, `len`, iteration and `.keys()` work from the file index; `.items()` / `.values()` and any
    mutation materialise every view first, after which this is an ordinary dict.
    """

    def __init__(self, files, counts, matrix, index=None):
        """index: the (seg list, file -> position, seg_off on the device) of an earlier table over the same files and
        counts (`.index`): a sweep embeds one list with many models, and the copy of seg_off from pageable memory
        is a blocking call queued behind the whole extraction on the current stream."""
        super().__init__()
        self.files = files
        self.counts = list(counts)
        self.matrix = matrix
        if index is None:
            seg = np.zeros(len(self.counts) + 1, dtype=np.int32)
            np.cumsum(np.asarray(self.counts, dtype=np.int64), out=seg[1:])
            index = (seg.tolist(),                     # segments of one file are contiguous rows
                     {f: i for i, f in enumerate(files)},   # a file listed twice: its last rows, as a dict built in order
                     torch.from_numpy(seg).to(matrix.device))
        self.index = index
        self._seg, self._where, self.seg_off = index[0], dict(index[1]), index[2]

    def _fill(self):
        if self._where is not None:
            where, self._where = self._where, None
            views = self.matrix.split(self.counts) if self.counts else ()
            for f, i in where.items():
                dict.__setitem__(self, f, views[i])

    def __getitem__(self, f):
        if self._where is None or dict.__contains__(self, f):
            return dict.__getitem__(self, f)
        i = self._where[f]                             # KeyError for an unknown file, as a dict
        v = self.matrix[self._seg[i]:self._seg[i + 1]]
        dict.__setitem__(self, f, v)
        return v

    def get(self, f, default=None):
        return self[f] if f in self else default

    def __contains__(self, f):
        return f in self._where if self._where is not None else dict.__contains__(self, f)

    def __len__(self):
        return len(self._where) if self._where is not None else dict.__len__(self)

    def __iter__(self):
        return iter(self._where) if self._where is not None else dict.__iter__(self)

    def keys(self):
        return self._where.keys() if self._where is not None else dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def copy(self):
        return dict(self.items())

    def __eq__(self, other):
        self._fill()
        return dict.__eq__(self, other)

    __hash__ = None

    def __reduce__(self):
        return (dict, (dict(self.items()),))

    def __repr__(self):
        return "VectorTable(%d files, %d x %d matrix on %s)" % (len(self), self.matrix.shape[0], self.matrix.shape[1],
                                                               self.matrix.device)

    def _mutating(name):
        def method(self, *args, **kwargs):
            self._fill()
            return getattr(dict, name)(self, *args, **kwargs)
        method.__name__ = name
        return method

    __setitem__ = _mutating("__setitem__")
    __delitem__ = _mutating("__delitem__")
    pop = _mutating("pop")
    popitem = _mutating("popitem")
    update = _mutating("update")
    setdefault = _mutating("setdefault")
    clear = _mutating("clear")
    del _mutating


class PackedUtterances:
    """
    An evaluation list staged ONCE on the GPU: utterances sorted by length, cut into batches (<= batch_size utterances,
    <= max_frames frames), each batch one flat float32 sample buffer with its index arrays.  extract_vectors accepts it
    in place of the dataset; a NAS sweep embeds the same list with every subnet and pays the host-side walk over the
    dataset (4 874 __getitem__ calls, ~45 ms) once instead of per subnet.
    """

    def __init__(self, datalst, device="cuda:0", batch_size=256, max_frames=262144, n_jobs=8, cache_features=False):
        """cache_features: keep the front-end's output (log-mel features, [frames, 80] in the model's compute dtype, a
        quarter of the waveforms' bytes) of every batch after the first extract_vectors call and feed later calls --
        other models over the same list -- from it: the front-end does not depend on the model, and in a NAS sweep it
        was a tenth of every sampled subnet's GPU time."""
        from .transforms import BatchPlan
        self.cache_features = bool(cache_features)
        self._features = {}                            # torch dtype -> list of [R, 80] tensors, one per batch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SugarError("sugar_amd runs on an MI355X only: pass device='cuda:N' (there is no CPU fallback)")
        self.table_index = None                        # VectorTable.index of the first table built over this list
        self.batches = []
        n = len(datalst)
        first = _segments(datalst[0])[0] if n else None
        if first is not None and not first.is_cuda:
            # a host list (files, or tensors in host memory): streamed through the pinned staging of _HostFeed -- items
            # read ahead by n_jobs threads, one native pack + ONE copy per batch -- and kept here batch by batch
            feed = _HostFeed(self.device, datalst, n_jobs, batch_size, max_frames)
            main = torch.cuda.current_stream(self.device)
            try:
                for k, slot, total, plan, rows in feed:
                    main.wait_event(feed.h2d_done[slot])
                    self.batches.append((rows, feed.dev[slot][:total].clone(), plan))
                    feed.done(slot, main)
            finally:
                feed.close()
            main.synchronize()                         # (rows / plans were allocated on the producer's thread: see extract_vectors)
            self.files, self.counts, self.n_rows = feed.files, feed.counts, feed.n_rows
            return
        files, counts, segs = [], [], []
        for idx in range(n):
            item = datalst[idx]
            files.append(item["file"])
            ss = _segments(item)
            counts.append(len(ss))
            segs.extend(ss)
        self.files, self.counts, self.n_rows = files, counts, len(segs)
        lengths = [int(s.numel()) for s in segs]
        for batch in plan_batches(lengths, max(1, batch_size), max_frames):
            flat = torch.cat([segs[i].reshape(-1).to(self.device, non_blocking=True).float() for i in batch])
            self.batches.append((torch.tensor(batch, device=self.device), flat, BatchPlan([lengths[i] for i in batch],
                                                                                          self.device)))

    def __len__(self):
        return len(self.files)

    def features(self, transform, dtype):
        """The cached front-end output of every batch in `dtype` (computed on the current stream at the first call)."""
        if dtype not in self._features:
            self._features[dtype] = [transform.rows(flat, plan, dtype) for _, flat, plan in self.batches]
        return self._features[dtype]

    def embed(self, spk, sub, k, normalize, feats=None):
        """Batch k through `sub` on the current stream: from the cached features when there are any, else the fused
        waveform -> embedding call."""
        idx, flat, plan = self.batches[k]
        if feats is not None:
            return sub.forward_rows(feats[k], plan.row_off, plan.B, plan.R, plan.t_max, normalize=normalize)
        return spk._embed_flat(sub, flat, plan, normalize)


def _packed_features(packed, spk, sub):
    """The feature cache of a packed list for this model, or None (cache off, train mode, or not the fused front-end)."""
    from .transforms import LogMelFbanks
    if not packed.cache_features or sub.training or not isinstance(getattr(spk, "__T__", None), LogMelFbanks):
        return None
    return packed.features(spk.__T__, torch.bfloat16 if sub.precision == "bf16" else torch.float32)


def _walk(datalst, n_jobs):
    """datalst[0], datalst[1], ... in order, loaded by `n_jobs` threads running ahead of the consumer (the reference's
    DataLoader workers / joblib threads, sugar/vectors.py:117-121, 130-133): reading and decoding a wav file releases the
    interpreter lock, and a serial walk over 4 874 files costs far more than embedding them.  The first items are
    loaded in line and timed: a list that is already in memory (items in tens of microseconds) is walked serially --
    a future per item costs more than such an item and the loader threads only fight the launching thread for the
    interpreter lock (4 874 in-memory items: 0.10 s serial, 0.26 s through the pool)."""
    import time
    n = len(datalst)
    probe = min(n, 8)
    t0 = time.perf_counter()
    for idx in range(probe):
        yield datalst[idx]          # (the consumer's time between two items is counted as well: an upper bound)
    if probe == n:
        return
    per_item = (time.perf_counter() - t0) / probe
    if n_jobs <= 1 or per_item < 2e-4:
        for idx in range(probe, n):
            yield datalst[idx]
        return
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    per_task = 4                    # items per future: amortises the pool's hand-over
    with ThreadPoolExecutor(max_workers=n_jobs, thread_name_prefix="sugar-load") as pool:
        window, nxt = deque(), probe
        while nxt < n or window:
            while nxt < n and len(window) < 4 * n_jobs:
                hi = min(n, nxt + per_task)
                window.append(pool.submit(lambda a, b: [datalst[i] for i in range(a, b)], nxt, hi))
                nxt = hi
            for item in window.popleft().result():
                yield item


class _HostFeed:
    """
    Host-fed batches for extract_vectors' list path and for PackedUtterances (the reference feeds its forward from
    DataLoader workers, sugar/vectors.py:126-153).  ONE producer thread streams the list: `workers` loader threads read
    items ahead in order (_walk); every CHUNK_FRAMES frames of loaded audio are sorted by length and cut into batches
    (<= batch_size utterances, <= max_frames frames); each batch is packed into a PINNED staging buffer with one
    native call (sugar_pack_segments: `workers` threads over equal byte ranges, no interpreter lock) and sent over PCIe
    on a copy stream while the GPU computes the batch before.  Three slots (pinned buffer + device buffer + two events
    each), so packing, the PCIe copy and the compute of three consecutive batches overlap; a slot is reused once the
    forward that read it has finished.  The list is never held whole: host memory is bounded by a chunk plus the slots
    whatever its length (Vox1-E / -H are 145 k / 138 k utterances), and the GPU starts on the first chunk while the
    rest is still being read.  Without this the list path walked the whole list first, serially, and then issued one
    synchronous pageable-memory copy per segment from the thread that also launches the kernels.

    Iterating yields (k, slot, n_samples, plan, rows): batch k sits in self.dev[slot][:n_samples] once
    self.h2d_done[slot] has fired; rows = the rows of the result matrix its utterances own (int64 on the device, list
    order).  After the iteration: self.files, self.counts, self.n_rows.
    """
    SLOTS = 3
    CHUNK_FRAMES = 1 << 20        # ~2.9 hours of audio: four full batches to sort and cut at a time

    def __init__(self, device, datalst, workers, batch_size, max_frames):
        import queue
        import threading
        self.device = device
        self.workers = max(1, int(workers))
        self.batch_size, self.max_frames = max(1, int(batch_size)), int(max_frames)
        self.pinned = [None] * self.SLOTS
        self.dev = [None] * self.SLOTS
        self.h2d_done = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.free = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.copy_stream = torch.cuda.Stream(device)
        self.q = queue.Queue()
        self.sem = threading.Semaphore(self.SLOTS)                # slots the consumer has handed back (see done())
        self._stop = False
        self.files, self.counts, self.n_rows = [], [], 0
        self.stats = {"load_s": 0.0, "pack_s": 0.0, "wait_slot_s": 0.0, "bytes": 0}   # producer-side seconds (diagnostics)
        self.thread = threading.Thread(target=self._produce, args=(datalst,), name="sugar-hostfeed", daemon=True)
        self.thread.start()

    # ---- producer thread -------------------------------------------------------------------------------------
    def _slot(self, s, need):
        """Staging buffers of slot s for `need` samples (grown when a batch -- one very long utterance -- exceeds them).
        (pin_memory=True: straight from torch's caching pinned allocator; `.pin_memory()` on a fresh tensor copies the
        whole buffer once more -- 0.5 GB of memcpy per call for three 168 MB slots.)"""
        if self.pinned[s] is None or self.pinned[s].numel() < need:
            cap = max(need, self.max_frames * 160)
            self.pinned[s] = torch.empty(cap, dtype=torch.float32, pin_memory=True)
            self.dev[s] = torch.empty(cap, dtype=torch.float32, device=self.device)
            # the caching allocator may hand out memory that kernels already queued on the allocating (default) stream
            # still use -- stream-ordered reuse; the copy stream writes into it, so it must run behind them
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))

    def _pack(self, dst, segs):
        """All float32 contiguous host segments of the batch in ONE native call; anything else with tensor.copy_."""
        import ctypes
        if all((not t.is_cuda) and t.dtype == torch.float32 and t.is_contiguous() for t in segs):
            n = len(segs)
            src = (ctypes.c_void_p * n)(*[t.data_ptr() for t in segs])
            lens = (ctypes.c_int64 * n)(*[t.numel() for t in segs])
            _lib.check(_lib.lib().sugar_pack_segments(src, lens, n, dst.data_ptr(), self.workers), "sugar_pack_segments")
            return
        off = 0
        for t in segs:
            dst[off:off + t.numel()].copy_(t.reshape(-1))
            off += t.numel()

    def _acquire(self, k):
        """Slot of batch k once the consumer has handed it back and the forward that read it has run; None: stopped."""
        import time
        t0 = time.perf_counter()
        self.sem.acquire()                                        # slot k % SLOTS: handed back by done() ...
        if self._stop:
            return None
        s = k % self.SLOTS
        if k >= self.SLOTS:
            self.free[s].synchronize()                            # ... and the forward that read it has finished
        self.stats["wait_slot_s"] += time.perf_counter() - t0
        return s

    def _emit(self, k, s, lengths, rows):
        """Batch k is in self.pinned[s]: its index arrays, the copy to the device, the hand-over to the consumer."""
        from .transforms import BatchPlan
        total = sum(lengths)
        self.stats["bytes"] += 4 * total
        plan = BatchPlan(lengths, self.device)
        idx = torch.tensor(rows, dtype=torch.int64).to(self.device)
        with torch.cuda.stream(self.copy_stream):
            self.dev[s][:total].copy_(self.pinned[s][:total], non_blocking=True)
            self.h2d_done[s].record(self.copy_stream)
        self.q.put((k, s, total, plan, idx))

    def _send(self, k, segs, rows):
        import time
        s = self._acquire(k)
        if s is None:
            return False
        t1 = time.perf_counter()
        lengths = [int(t.numel()) for t in segs]
        self._slot(s, sum(lengths))
        self._pack(self.pinned[s], segs)
        self.stats["pack_s"] += time.perf_counter() - t1
        self._emit(k, s, lengths, rows)
        return True

    def _native_list(self, datalst):
        """An evaluation list of wav FILES (database.Utterance in eval mode, possibly a rank's _Subset of one) whose
        files are all mono 16-bit PCM at 16 kHz: (paths, samples per file, data offsets, segment starts [n, n_seg],
        segment lengths [n]) for the native reader (sugar_wav_probe / sugar_wav_load), else None -- any other dataset or
        format goes through its own __getitem__."""
        import ctypes
        from .database import Utterance
        from .waveforms import eval_segments
        base, order = (datalst.data, datalst.indices) if isinstance(datalst, _Subset) else (datalst, None)
        if type(base) is not Utterance or not base.mode_eval or base.ndim != 1 or len(datalst) == 0:
            return None
        paths = [base.path(i) for i in (order if order is not None else range(len(base)))]
        n = len(paths)
        arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
        size = np.empty(n, dtype=np.int64)
        off = np.empty(n, dtype=np.int64)
        rc = _lib.lib().sugar_wav_probe(arr, n, 16000, size.ctypes.data, off.ctypes.data, self.workers)
        if rc != 0 or (size < 0).any():
            return None                                # (unreadable files raise from the Python loader, with its message)
        starts, lengths = eval_segments(size, base.num_samples, base.num_eval)
        return paths, arr, size, off, starts, lengths

    def _produce_native(self, native):
        import time
        paths, arr, size, off, starts, lengths = native
        n, n_seg = starts.shape
        self.files, self.counts, self.n_rows = list(paths), [n_seg] * n, n * n_seg
        seg_frames = 1 + lengths // 160                # num_frames of every segment of a file
        k, lo = 0, 0
        while lo < n:
            hi, frames = lo, 0
            while hi < n and frames < self.CHUNK_FRAMES:
                frames += int(seg_frames[hi]) * n_seg
                hi += 1
            # the chunk's segments, file-major: segment j of file f is row f * n_seg + j of the result
            f_idx = np.repeat(np.arange(lo, hi, dtype=np.int32), n_seg)
            s_start = starts[lo:hi].reshape(-1)
            s_len = np.repeat(lengths[lo:hi], n_seg)
            rows = np.arange(lo * n_seg, hi * n_seg, dtype=np.int64)
            for batch in plan_batches(s_len.tolist(), self.batch_size, self.max_frames):
                s = self._acquire(k)
                if s is None:
                    return
                t1 = time.perf_counter()
                b = np.asarray(batch, dtype=np.int64)
                bf, bs, bl = np.ascontiguousarray(f_idx[b]), np.ascontiguousarray(s_start[b]), np.ascontiguousarray(s_len[b])
                self._slot(s, int(bl.sum()))
                _lib.check(_lib.lib().sugar_wav_load(arr, size.ctypes.data, off.ctypes.data, n, bf.ctypes.data, bs.ctypes.data,
                                                     bl.ctypes.data, len(batch), self.pinned[s].data_ptr(), self.workers),
                           "sugar_wav_load")
                self.stats["load_s"] += time.perf_counter() - t1
                self._emit(k, s, bl.tolist(), rows[b].tolist())
                k += 1
            lo = hi

    def _produce(self, datalst):
        import time
        try:
            with torch.cuda.device(self.device):
                native = self._native_list(datalst)
                if native is not None:
                    self.stats["native_reader"] = 1
                    self._produce_native(native)
                    if not self._stop:
                        self.q.put(None)
                    return
                k, row, segs, rows, frames = 0, 0, [], [], 0

                def flush():
                    nonlocal k, segs, rows, frames
                    lengths = [int(t.numel()) for t in segs]
                    for batch in plan_batches(lengths, self.batch_size, self.max_frames):
                        if not self._send(k, [segs[i] for i in batch], [rows[i] for i in batch]):
                            return False
                        k += 1
                    segs, rows, frames = [], [], 0
                    return True

                t0 = time.perf_counter()
                for item in _walk(datalst, self.workers):
                    ss = _segments(item)
                    self.files.append(item["file"])
                    self.counts.append(len(ss))
                    for t in ss:
                        segs.append(t)
                        rows.append(row)
                        row += 1
                        frames += num_frames(int(t.numel()))
                    if frames >= self.CHUNK_FRAMES:
                        self.stats["load_s"] += time.perf_counter() - t0
                        if not flush():
                            return
                        t0 = time.perf_counter()
                    if self._stop:
                        return
                self.stats["load_s"] += time.perf_counter() - t0
                self.n_rows = row
                if segs and not flush():
                    return
            self.q.put(None)
        except BaseException as e:                                # surfaces in the consumer
            self.q.put(e)

    # ---- consumer --------------------------------------------------------------------------------------------
    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def done(self, slot, stream):
        """The forward reading `slot` has been enqueued on `stream`: the slot may be refilled once it has run."""
        self.free[slot].record(stream)
        self.sem.release()

    def close(self):
        self._stop = True
        self.sem.release()                                        # a producer waiting for a slot sees _stop
        self.thread.join()


def _dist_info(rank, world_size, shard):
    """
    (rank, world) of this call.  Sharding is OPT-IN: an explicit ``world_size`` (with ``rank``, default: the process
    group's), or ``shard=True`` (both from the initialised process group).  A process group that merely exists does NOT
    shard: under the reference's multi-GPU evaluation every rank holds its OWN checkpoint / subnet and embeds the whole
    list with no collective (scripts/train_vox2_veri.py:385-403 strides CHECKPOINTS over the GPUs, not utterances).
    """
    import torch.distributed as dist
    live = dist.is_available() and dist.is_initialized()
    if world_size is None and not shard:
        return 0, 1
    if world_size is None:
        if not live:
            return 0, 1
        world_size = dist.get_world_size()
    if rank is None:
        rank = dist.get_rank() if live else 0
    rank, world_size = int(rank), int(world_size)
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("extract_vectors: rank %d outside world_size %d" % (rank, world_size))
    return rank, world_size


class _Subset:
    """The items `indices` of a dataset, in that order (a rank's shard)."""

    def __init__(self, data, indices):
        self.data, self.indices = data, indices

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        return self.data[self.indices[i]]


def extract_vectors(model, datalst, norm=True, device="cuda:0", n_jobs=4, batch_size=1, max_frames=262144,
                    rank=None, world_size=None, shard=False):
    """
    model   : WrappedModel / SpeakerModel from sugar_amd.models
    datalst : an indexable dataset of {"file": str, "input": Tensor[n_seg, L] or [L]} (sugar/database.py:136-142),
              or a PackedUtterances built from one (then `device`, `batch_size`, `max_frames` are the packing's)
    batch_size : utterances per packed call, at most (and at most `max_frames` frames).  1 -- the reference's default and
              its only choice for whole utterances of different lengths, which it then runs ONE AT A TIME
              (sugar/vectors.py:98-124) -- means "whole utterances" here as well, not "one per call": ragged utterances
              pack into one call without padding, so the packer takes up to AUTO_BATCH (256) of them.  An utterance's
              vector does not depend on what it is batched with beyond the rounding of the compute dtype.
    n_jobs  : host threads that pack the next batch into pinned staging memory while the GPU computes (the reference's
              joblib threads / DataLoader workers, sugar/vectors.py:117-121, 130-133); only used for host-resident lists
    rank, world_size, shard : multi-GPU extraction of ONE model's vectors (BASELINE config 4) -- opt-in, never
              inferred from a process group that happens to exist: pass ``world_size`` (and ``rank``), or
              ``shard=True`` to take both from the initialised torch.distributed group.  ALL ranks must then call with
              the SAME model and list: every rank embeds a shard (items rank, rank + N, ...; longest-first round-robin
              when the dataset exposes `lengths`), the shards are exchanged with one all-gather of the [n_local, 192]
              blocks (RCCL over xGMI; gloo in tests) and EVERY rank returns the complete table, rows and files in the
              order of `datalst` -- bit-identical to the one-rank result.  The default (no sharding, no collective) is
              the reference's behaviour under a live group: each rank evaluates its own checkpoint over the whole list
              (scripts/train_vox2_veri.py:385-403), which is also what `evaluate_subnets` and `veri_validate` do.
    Returns a VectorTable: {file: Tensor[n_seg, 192]} on the device.
    """
    rank, world = _dist_info(rank, world_size, shard)
    batch_size = AUTO_BATCH if int(batch_size) <= 1 else int(batch_size)
    if world > 1:
        return _extract_sharded(model, datalst, norm, device, n_jobs, batch_size, max_frames, rank, world)
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
    lanes = stream_lanes(device)
    for lane in lanes:
        lane.wait_stream(main)
    if packed is not None:
        files, counts = packed.files, packed.counts
        out = torch.empty(packed.n_rows, 192, dtype=torch.float32, device=device)
        sub = spk._subnet()
        feats = _packed_features(packed, spk, sub)     # (computed on `main` at the first call: before the lanes wait for it)
        if feats is not None:
            for lane in lanes:
                lane.wait_stream(main)
        for k, (idx, flat, plan) in enumerate(packed.batches):
            with torch.cuda.stream(lanes[k % len(lanes)]):
                out[idx] = packed.embed(spk, sub, k, bool(norm), feats)
    elif len(datalst) and not _segments(datalst[0])[0].is_cuda:
        # host-resident list (files, or tensors in host memory): streamed -- items read ahead by n_jobs threads, pinned
        # triple-buffered staging on a copy stream, the GPU starts on the first chunk (_HostFeed)
        from .transforms import LogMelFbanks
        if not isinstance(getattr(spk, "__T__", None), LogMelFbanks):
            raise _lib.SugarError("extract_vectors needs the fused LogMelFbanks + TDNN path")
        sub = spk._subnet()
        sub.eval_handle(device)                                   # pack the weights before the pipeline starts
        feed = _HostFeed(device, datalst, max(1, int(n_jobs)), batch_size, max_frames)
        # the batch plans and row indices are allocated by the producer thread (on ITS stream) and read by kernels
        # on the lanes: they stay referenced until the lanes have been joined, or the caching allocator would hand
        # their memory to the producer's next batch while a lane still has to read it
        keep, embs = [], []
        try:
            for k, slot, total, plan, rows in feed:
                lane = lanes[k % len(lanes)]
                lane.wait_event(feed.h2d_done[slot])
                with torch.cuda.stream(lane):
                    emb = spk._embed_flat(sub, feed.dev[slot][:total], plan, bool(norm))
                keep.append((plan, rows, emb))
                embs.append((rows, emb))
                feed.done(slot, lane)
        finally:
            feed.close()
        extract_vectors.last_hostfeed_stats = dict(feed.stats)       # diagnostics (tools/hostfed_bench.py)
        for lane in lanes:
            lane.synchronize()
        files, counts = feed.files, feed.counts
        out = torch.empty(feed.n_rows, 192, dtype=torch.float32, device=device)
        if embs:
            out[torch.cat([r for r, _ in embs])] = torch.cat([e for _, e in embs])
        del keep, embs
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
            with torch.cuda.stream(lanes[k % len(lanes)]):
                wavs = [segs[i].to(device, non_blocking=True) for i in batch]
                emb = spk.embed_ragged(wavs, normalize=bool(norm))
                out[torch.tensor(batch, device=device)] = emb
    for lane in lanes:
        main.wait_stream(lane)
    if packed is None:
        return VectorTable(files, counts, out)
    table = VectorTable(files, counts, out, index=packed.table_index)
    packed.table_index = table.index
    return table


def _extract_sharded(model, datalst, norm, device, n_jobs, batch_size, max_frames, rank, world):
    """extract_vectors on `world` ranks: shard -> embed -> exchange -> the complete table on every rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() != world:
        raise _lib.SugarError("extract_vectors(world_size=%d) needs an initialised torch.distributed process group of "
                              "that size (backend 'nccl' = RCCL on the GPUs)" % world)
    if isinstance(datalst, PackedUtterances):
        # the list is already on this rank's GPU in length-sorted batches: deal the batches round-robin (neighbouring
        # batches have neighbouring lengths, so the shards balance); every rank knows every rank's rows, so the
        # exchange is ONE all_gather_into_tensor and nothing else
        packed, device = datalst, datalst.device
        model.to(device)
        model.eval()
        spk = model.module if hasattr(model, "module") else model
        sub = spk._subnet()
        main = torch.cuda.current_stream(device)
        lanes = stream_lanes(device)
        for lane in lanes:
            lane.wait_stream(main)
        share = [[k for k in range(len(packed.batches)) if k % world == r] for r in range(world)]
        parts = []
        for j, k in enumerate(share[rank]):
            with torch.cuda.stream(lanes[j % len(lanes)]):
                parts.append(packed.embed(spk, sub, k, bool(norm)))
        for lane in lanes:
            main.wait_stream(lane)
        local = torch.cat(parts) if parts else torch.empty(0, 192, dtype=torch.float32, device=device)
        sizes = [sum(int(packed.batches[k][0].numel()) for k in ks) for ks in share]
        blocks = gather_vectors(local, world, sizes=sizes)
        out = torch.empty(packed.n_rows, 192, dtype=torch.float32, device=device)
        for r in range(world):
            if share[r]:
                out[torch.cat([packed.batches[k][0] for k in share[r]])] = blocks[r]
        table = VectorTable(packed.files, packed.counts, out, index=packed.table_index)
        packed.table_index = table.index
        return table
    n = len(datalst)
    mine = shard_indices(n, rank, world, getattr(datalst, "lengths", None))
    local = extract_vectors(model, _Subset(datalst, mine), norm, device, n_jobs, batch_size, max_frames)
    # two collectives: one object gather of the host-side bookkeeping (item indices, names, segment counts -- which
    # also gives every rank's row count), one all_gather_into_tensor of the vectors
    meta = [None] * world
    with _collective_device(local.matrix):
        dist.all_gather_object(meta, (mine, local.files, local.counts))
    blocks = gather_vectors(local.matrix, world, sizes=[sum(m[2]) for m in meta])
    return VectorTable(*assemble_shards(n, meta, blocks, local.matrix.device))


def _collective_device(t):
    """
    Context for a torch.distributed call next to tensor `t`: under backend "nccl" (RCCL) object collectives serialise
    through `torch.cuda.current_device()` and tensor collectives pick their communicator by it, so the current device
    must be the rank's own GPU -- which a caller that only passed device="cuda:N" has not set.
    """
    import contextlib
    return torch.cuda.device(t.device) if t.is_cuda else contextlib.nullcontext()


def assemble_shards(n, meta, blocks, device):
    """
    Put gathered shards back into list order.  meta[r] = (item indices, files, segment counts) of rank r's shard,
    blocks[r] = its [rows, D] vectors (items in that order, an item's segments contiguous).  Returns (files, counts,
    matrix) for VectorTable: item i of the original list owns rows seg_off[i] .. seg_off[i+1] of `matrix`.
    """
    files, counts, src = [None] * n, [0] * n, [None] * n
    for r, (idx, fl, ct) in enumerate(meta):
        off = 0
        for i, f, c in zip(idx, fl, ct):
            files[i], counts[i], src[i] = f, c, (r, off)
            off += c
    if any(f is None for f in files):
        raise _lib.SugarError("extract_vectors: the ranks' shards do not cover the list")
    base, acc = [], 0
    for b in blocks:
        base.append(acc)
        acc += b.shape[0]
    perm = [base[r] + off + j for (r, off), c in zip(src, counts) for j in range(c)]
    matrix = torch.cat([b.to(device) for b in blocks])[torch.tensor(perm, dtype=torch.int64, device=device)]
    return files, counts, matrix


def gather_vectors(local, world_size=None, sizes=None):
    """
    Multi-GPU exchange step: every rank embedded a shard of the utterances; ONE all_gather_into_tensor of
    the [n_local, 192] float32 blocks (RCCL over xGMI on GPUs, gloo in CPU tests) gives every rank all vectors.
    ``local``: Tensor [n_local, D]; shards may differ in size (padded to the max for the collective).
    ``sizes``: every rank's row count when the caller knows them (extract_vectors does); else they are exchanged first
    with one all_gather_object.
    Returns the list of per-rank tensors in rank order (on ``local``'s device).
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    world = dist.get_world_size() if world_size is None else world_size
    with _collective_device(local):
        if sizes is None:
            sizes = [None] * world
            dist.all_gather_object(sizes, int(local.shape[0]))
        sizes = [int(x) for x in sizes]
        if len(sizes) != world or sizes[dist.get_rank()] != local.shape[0]:
            raise _lib.SugarError("gather_vectors: sizes %r do not match this rank's %d rows" % (sizes, local.shape[0]))
        # a gloo group moves host memory: stage device tensors through the host (tests; two ranks sharing one GPU)
        wire = local.cpu() if (local.is_cuda and dist.get_backend() == "gloo") else local
        m = max(sizes)
        pad = wire if wire.shape[0] == m else torch.cat([wire, wire.new_zeros(m - wire.shape[0], wire.shape[1])])
        buf = wire.new_empty(world * m, wire.shape[1])
        dist.all_gather_into_tensor(buf, pad.contiguous())
    return [buf[r * m:r * m + sizes[r]].to(local.device) for r in range(world)]


def shard_indices(n_items, rank, world, weights=None):
    """Deal items to ranks: longest-first round-robin when weights (lengths) are given, else strided."""
    if weights is None:
        return list(range(rank, n_items, world))
    order = sorted(range(n_items), key=lambda i: -weights[i])
    return sorted(order[rank::world])


def length_normalize(ivectors, scaleup=False):
    """
    sugar/vectors.py:25-60 (host helper, torch tensor or ndarray in -> same kind out): vectors divided by their 2-norm
    (by norm / sqrt(dim) with `scaleup`), plus the mean and the population std of that ratio.
    """
    import numpy as np
    if isinstance(ivectors, torch.Tensor):
        ratio = ivectors.norm(p=2, dim=-1, keepdim=True)
    elif isinstance(ivectors, np.ndarray):
        ratio = np.linalg.norm(ivectors, ord=2, axis=-1, keepdims=True)
    else:
        raise TypeError("length_normalize expects a torch.Tensor or a numpy.ndarray")
    if scaleup is True:
        ratio = ratio / float(np.sqrt(ivectors.shape[-1]))
    ratio_avg = ratio.mean()
    ratio_std = ((ratio ** 2).mean() - ratio_avg ** 2) ** 0.5
    return ivectors / ratio, ratio_avg, ratio_std


def vector_transform(vectors, transform=None):
    """sugar/vectors.py:157-178: {key: transform(value)}, or a shallow copy of the dict when transform is None."""
    if transform is None:
        return {k: v for k, v in vectors.items()}
    return {k: transform(v) for k, v in vectors.items()}
