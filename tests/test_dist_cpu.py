"""CPU, world_size 2 over gloo: the multi-GPU exchange step (utterance shards + one all-gather of embeddings)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sugar_amd import synth
from sugar_amd.vectors import assemble_shards, gather_vectors, shard_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_utts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths = synth.ragged_lengths(n_utts, seed=5).tolist()
    mine = shard_indices(n_utts, rank, world, lengths)
    # stand-in for the embeddings this rank computed: row u of the global table
    table = synth.normal("dist/emb", (n_utts, 192), seed=5)
    local = table[mine]
    parts = gather_vectors(local)
    # the same exchange when the caller knows every rank's row count: one collective, same blocks
    known = [len(shard_indices(n_utts, r, world, lengths)) for r in range(world)]
    parts2 = gather_vectors(local, world, sizes=known)
    assert all(torch.equal(a, b) for a, b in zip(parts, parts2))
    # sharding is opt-in: a live process group alone does not shard (evaluate_subnets / veri_validate rely on it)
    from sugar_amd.vectors import _dist_info
    assert _dist_info(None, None, False) == (0, 1)
    assert _dist_info(None, None, True) == (rank, world)
    assert _dist_info(None, world, False) == (rank, world)
    assert _dist_info(1, 4, False) == (1, 4)
    full = torch.empty(n_utts, 192)
    for r, part in enumerate(parts):
        full[shard_indices(n_utts, r, world, lengths)] = part
    q.put((rank, bool(torch.equal(full, table)), [p.shape[0] for p in parts]))
    dist.destroy_process_group()


def test_allgather_uneven_shards_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_utts = 37   # odd on purpose: shards of 19 and 18 rows
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_utts, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, sizes in res:
        assert ok, rank
        assert sizes == [19, 18]


def _assemble_worker(rank, world, port, n_utts, q):
    """The exchange + reassembly of the sharded extract_vectors (vectors._extract_sharded) on stand-in vectors:
    strided shards as in the reference (train_vox2_veri.py:385-403), 1-3 segments per utterance."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counts = [1 + (i * 7) % 3 for i in range(n_utts)]
    seg_off = [0]
    for c in counts:
        seg_off.append(seg_off[-1] + c)
    table = synth.normal("dist/segs", (seg_off[-1], 192), seed=6)
    mine = shard_indices(n_utts, rank, world)
    local = torch.cat([table[seg_off[i]:seg_off[i + 1]] for i in mine])
    meta = [None] * world
    dist.all_gather_object(meta, (mine, ["utt%03d" % i for i in mine], [counts[i] for i in mine]))
    files, got_counts, matrix = assemble_shards(n_utts, meta, gather_vectors(local, world), torch.device("cpu"))
    ok = files == ["utt%03d" % i for i in range(n_utts)] and got_counts == counts and torch.equal(matrix, table)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharded_extraction_reassembles_list_order_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_assemble_worker, args=(r, 2, port, 23, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_gather_without_process_group_is_identity():
    x = torch.ones(3, 192)
    assert gather_vectors(x)[0] is x


# ---- NAS sweep (BASELINE config 5): configs dealt to ranks by multiply-adds, results gathered once ------------------
def _sweep_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sugar_amd
    from sugar_amd.sweep import config_key, gather_results, shard_configs
    sup = sugar_amd.tdnn8m2g()
    path = sugar_amd.TDNNPath(sup.space(mode=["kernel", "depth", "width1", "width2"]))
    configs = [config_key(path.sample()) for _ in range(11)]
    before = sup.config
    mine = shard_configs(sup, configs, rank, world)
    assert sup.config == before                      # costing the configs leaves the active config alone
    # stand-in for the evaluation this rank ran on its GPU
    local = {configs[i]: {"eer": float(i), "rank": rank} for i in mine}
    full = gather_results(local)
    q.put((rank, mine, sorted(full.items())))
    dist.destroy_process_group()


def test_sweep_shards_and_gathers_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sweep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, full0), (r1, mine1, full1) = res
    assert sorted(mine0 + mine1) == list(range(11)) and abs(len(mine0) - len(mine1)) <= 1
    assert full0 == full1 and len(full0) == len({k for k, _ in full0})
    for _, v in full0:
        assert int(v["eer"]) in (mine0 if v["rank"] == 0 else mine1)


def test_sharding_is_opt_in_without_a_group():
    from sugar_amd.vectors import _dist_info
    assert _dist_info(None, None, False) == (0, 1)
    assert _dist_info(None, None, True) == (0, 1)          # shard=True without a process group: one rank
    assert _dist_info(None, 1, False) == (0, 1)
    import pytest
    with pytest.raises(ValueError):
        _dist_info(3, 2, False)
