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


def _run(sugar_amd, model, data, trials):
    out = {}
    for name, lst in (("list", data), ("packed", sugar_amd.PackedUtterances(data, DEV, batch_size=4))):
        table = sugar_amd.extract_vectors(model, lst, norm=True, device=DEV, batch_size=4)
        scores, _, _ = sugar_amd.score_trials(trials, table)
        torch.cuda.synchronize()
        out[name] = (list(table.files), table.seg_off.cpu().numpy(), table.matrix.cpu().numpy(), scores)
    return out


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _run(*_setup())))
    finally:
        dist.destroy_process_group()


def test_two_rank_extraction_and_scoring_is_bit_equal_to_one_rank():
    want = _run(*_setup())                      # no process group in this process: the one-rank path
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in (0, 1):                         # every rank holds the complete table
        for name in ("list", "packed"):
            files, seg, mat, scores = res[rank][name]
            wf, ws, wm, wsc = want[name]
            assert files == wf and np.array_equal(seg, ws), (rank, name)
            assert np.array_equal(mat, wm), (rank, name, np.abs(mat - wm).max())
            assert np.array_equal(scores, wsc), (rank, name)
    assert np.array_equal(want["list"][2], want["packed"][2])
