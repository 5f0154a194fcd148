"""
GPU: BASELINE config 4 end to end on TWO ranks -- shard the utterance list, embed the shard, ONE all-gather of the
[n_local, 192] blocks, the complete table on every rank, cosine trial scoring.  Both ranks compute on cuda:0 (the test
box has one GPU) and exchange over gloo through host memory; on a node the same code runs one rank per GPU with backend
"nccl" (= RCCL over xGMI).  Scores must be BIT-equal to the one-rank run: an utterance's embedding does not depend on
what else is in its batch, so sharding may not change a single bit.
Reference behaviour kept: extract_vectors' return layout (sugar/vectors.py:63-154) and the rank striding of
scripts/train_vox2_veri.py:385-403.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_UTTS, N_TRIALS = 26, 300


class _Utts:
    def __init__(self, wavs, segs):
        self.wavs, self.segs = wavs, segs

    def __len__(self):
        return len(self.wavs)

    def __getitem__(self, i):
        w, n = self.wavs[i], self.segs[i]
        L = w.numel() // n
        return {"file": "utt%04d.wav" % i, "input": w[:n * L].view(n, L) if n > 1 else w.unsqueeze(0)}


def _setup():
    import sugar_amd
    from sugar_amd import layout, synth
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    sub = sup.clone(layout.SMALLEST).set_precision("bf16")
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()
    lens = synth.ragged_lengths(N_UTTS, lo=32000, hi=120000, seed=17).tolist()
    data = _Utts(synth.ragged_waveforms(lens, seed=17), [1 + (i % 3 == 0) for i in range(N_UTTS)])
    lab, a, b = synth.trial_pairs(N_UTTS, N_TRIALS, seed=17)
    trials = [{"label": int(lab[i]), "enroll": "utt%04d.wav" % a[i], "test": "utt%04d.wav" % b[i]} for i in range(N_TRIALS)]
    return sugar_amd, model, data, trials


def _run(sugar_amd, model, data, trials, shard=False):
    out = {}
    for name, lst in (("list", data), ("packed", sugar_amd.PackedUtterances(data, DEV, batch_size=4))):
        table = sugar_amd.extract_vectors(model, lst, norm=True, device=DEV, batch_size=4, shard=shard)
        scores, _, _ = sugar_amd.score_trials(trials, table)
        torch.cuda.synchronize()
        out[name] = (list(table.files), table.seg_off.cpu().numpy(), table.matrix.cpu().numpy(), scores)
    return out


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _run(*_setup(), shard=True)))      # sharding is opt-in
    finally:
        dist.destroy_process_group()


def _two_ranks(worker, *extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q) + extra) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_two_rank_extraction_and_scoring_is_bit_equal_to_one_rank():
    want = _run(*_setup())                      # no process group in this process: the one-rank path
    res = _two_ranks(_worker)
    for rank in (0, 1):                         # every rank holds the complete table
        for name in ("list", "packed"):
            files, seg, mat, scores = res[rank][name]
            wf, ws, wm, wsc = want[name]
            assert files == wf and np.array_equal(seg, ws), (rank, name)
            assert np.array_equal(mat, wm), (rank, name, np.abs(mat - wm).max())
            assert np.array_equal(scores, wsc), (rank, name)
    assert np.array_equal(want["list"][2], want["packed"][2])


# ---- by-MODEL split under a live process group: every rank its own subnets over the WHOLE list, no collective -----------
# (BASELINE config 5; the reference's multi-GPU evaluation, scripts/train_vox2_veri.py:385-403: under
# init_process_group rank g loads checkpoints g, g + N, ... and calls veri_validate on each -- no collective anywhere;
# SubnetEvaluation.ipynb cell ec189126 is the per-config loop)
N_SWEEP_UTTS, N_SWEEP_TRIALS = 12, 120


class _Trials(list):
    utterances = None


class _Loader:
    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size, self.num_workers = dataset, batch_size, 0


def _sweep_setup():
    import sugar_amd
    from sugar_amd import layout, synth
    from sugar_amd.sweep import config_key
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    path = sugar_amd.TDNNPath(sup.space(mode=["kernel", "depth", "width1", "width2"], verbose=False))
    configs = [config_key(path.sample()) for _ in range(5)]          # 5 configs on 2 ranks: 3 / 2, uneven on purpose
    lens = synth.ragged_lengths(N_SWEEP_UTTS, lo=24000, hi=40000, seed=13).tolist()
    lab, a, b = synth.trial_pairs(N_SWEEP_UTTS, N_SWEEP_TRIALS, seed=13)
    trials = _Trials({"label": int(lab[i]), "enroll": "utt%04d.wav" % a[i], "test": "utt%04d.wav" % b[i]}
                     for i in range(len(lab)))
    trials.utterances = _Utts(synth.ragged_waveforms(lens, seed=13), [1] * N_SWEEP_UTTS)
    recal = [(synth.waveforms(6, 32000, seed=40 + i),) for i in range(2)]
    return sugar_amd, sup, configs, _Loader(trials, 4), recal


def _strip(results):
    return {k: {f: v for f, v in r.items() if f != "rank"} for k, r in results.items()}


def _sweep_run(sugar_amd, sup, configs, loader, recal):
    return {name: sugar_amd.evaluate_subnets(sup, configs, loader, recal_loader=rl, device=DEV)
            for name, rl in (("plain", None), ("recal", recal))}


def _sweep_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _sweep_run(*_sweep_setup())))
    finally:
        dist.destroy_process_group()


def test_evaluate_subnets_two_ranks_uneven_share_equals_one_rank():
    """5 configs dealt 3 / 2 to two ranks under a live process group, with and without BN recalibration: every rank ends
    with the results of ALL configs, bit-equal to the one-rank sweep (round 2 sharded the utterance list inside each
    rank's extract_vectors and all-gathered embeddings of different subnets -- or hung on the uneven share)."""
    want = _sweep_run(*_sweep_setup())
    res = _two_ranks(_sweep_worker)
    for name in ("plain", "recal"):
        w = _strip(want[name])
        assert len(w) == 5
        for rank in (0, 1):
            got = res[rank][name]
            assert _strip(got) == w, (name, rank)
            owners = sorted(r["rank"] for r in got.values())
            assert owners in ([0, 0, 0, 1, 1], [0, 0, 1, 1, 1]), owners      # really split over the ranks


def _own_model_worker(rank, world, port, q):
    """The reference pattern: process group live, rank g evaluates ITS OWN models g, g + N, ... with veri_validate."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sugar_amd, sup, configs, loader, _ = _sweep_setup()
        configs = configs[:3]                                          # 2 models on rank 0, 1 on rank 1

        def forbidden(*a, **k):
            raise AssertionError("collective on the by-model path")
        saved = {n: getattr(dist, n) for n in ("all_gather", "all_gather_object", "all_gather_into_tensor", "all_reduce",
                                                "broadcast", "barrier")}
        out = {}
        try:
            for n in saved:
                setattr(dist, n, forbidden)
            for i in range(rank, len(configs), world):
                sub = sup.clone(configs[i]).set_precision("bf16")
                model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()
                eer, dcf, vectors, scs = sugar_amd.veri_validate(loader, model, device=DEV, ret_info=True)
                out[i] = (eer, dcf, vectors.matrix.cpu().numpy(), scs)
        finally:
            for n, f in saved.items():
                setattr(dist, n, f)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_veri_validate_per_rank_model_under_live_group_has_no_collective():
    sugar_amd, sup, configs, loader, _ = _sweep_setup()
    res = _two_ranks(_own_model_worker)
    assert sorted(res[0]) == [0, 2] and sorted(res[1]) == [1]
    for i, cfg in enumerate(configs[:3]):
        sub = sup.clone(cfg).set_precision("bf16")
        model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(DEV).eval()
        eer, dcf, vectors, scs = sugar_amd.veri_validate(loader, model, device=DEV, ret_info=True)
        g_eer, g_dcf, g_mat, g_scs = res[i % 2][i]
        assert g_eer == eer and g_dcf == dcf, (i, g_eer, eer)
        assert np.array_equal(g_mat, vectors.matrix.cpu().numpy()) and np.array_equal(g_scs, scs)
