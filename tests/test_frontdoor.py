"""
Data front door (loadWAV, Utterance, VerificationTrials; sugar/waveforms.py:47-115, sugar/database.py:71-158, 289-364;
SURVEY 8f rank 3) against what the reference's own functions returned on the same seeded files
(tests/golden/frontdoor.json, made by oracle/make_golden_frontdoor.py; arrays compared exactly by SHA-256).
"""
import hashlib
import json
import os
import random

import numpy as np
import pytest
import torch

import sugar_amd
from oracle.make_golden_frontdoor_inputs import LOAD_CASES, WAVS, write_files
from sugar_amd import layout, synth
from sugar_amd.database import Utterance, VerificationTrials, txt2list
from sugar_amd.waveforms import loadWAV, load_wav

with open(os.path.join(os.path.dirname(__file__), "golden", "frontdoor.json")) as f:
    GOLD = json.load(f)


def _same(arr, key):
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    shape, digest = GOLD["arrays"][key]
    assert list(arr.shape) == shape, (key, arr.shape, shape)
    assert hashlib.sha256(arr.tobytes()).hexdigest() == digest, key


def test_loadwav_matches_reference(tmp_path):
    root = str(tmp_path)
    write_files(root)
    for ci, (ns, ev, ne, mn, seed) in enumerate(LOAD_CASES):
        for name in WAVS:
            random.seed(seed)
            w = loadWAV(os.path.join(root, name), ns, ev, ne, 16000, mn)
            assert w.dtype == np.float32
            _same(w, "load/%d/%s" % (ci, name))
    assert load_wav(os.path.join(root, "spk1/a.wav")).shape == (20000,)
    with pytest.raises(AssertionError):
        loadWAV(os.path.join(root, "spk1/a.wav"), sample_rate=8000)


def test_trials_and_utterances_match_reference(tmp_path):
    root = str(tmp_path)
    write_files(root)
    rel = lambda p: os.path.relpath(p, root)
    random.seed(5)
    tr = VerificationTrials(os.path.join(root, "trials.txt"), num_samples=0, mode_eval=True, num_eval=1, trials_dir=root)
    assert [int(x) for x in tr.labels] == GOLD["labels"]          # the third line has no label: random.randint
    assert [rel(p) for p in tr.enrolls] == GOLD["enrolls"] and [rel(p) for p in tr.tests] == GOLD["tests"]
    assert [rel(p) for p in tr.files] == GOLD["files"] == [rel(p) for p in tr.trials2utts()]
    assert len(tr) == 5
    it = tr[2]
    assert {"label": int(it["label"]), "enroll": rel(it["enroll"]), "test": rel(it["test"])} == GOLD["item2"]
    u0 = tr.utterances[0]
    assert rel(u0["file"]) == GOLD["utt0_file"]
    _same(u0["input"].numpy(), "utt0_input")
    ds = Utterance(list(WAVS), num_samples=16000, mode_eval=True, num_eval=3, root_dir=root)
    batch = ds.collate_fn([ds[0], ds[2]])
    assert [rel(p) for p in batch["file"]] == GOLD["collate_files"] and batch["num"] == GOLD["collate_num"]
    _same(batch["input"].numpy(), "collate_input")
    assert [rel(p) for p in txt2list(os.path.join(root, "utts.txt"), root)] == GOLD["txt2list"]


class _Loader:
    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size, self.num_workers = dataset, batch_size, 0


@pytest.mark.gpu
def test_veri_validate_from_files(tmp_path):
    """Trial list + wav files -> VerificationTrials -> veri_validate on the GPU; whole-utterance vectors
    (num_samples = 0) of different lengths run as one packed batch."""
    root = str(tmp_path)
    write_files(root)
    random.seed(5)
    tr = VerificationTrials(os.path.join(root, "trials.txt"), num_samples=0, mode_eval=True, num_eval=1, trials_dir=root)
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    sub = sup.clone(layout.SMALLEST).set_precision("fp32")
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).cuda().eval()
    eer, dcf, vectors, scs = sugar_amd.veri_validate(_Loader(tr, 4), model, device="cuda:0", ret_info=True)
    assert sorted(vectors) == tr.files and scs.shape == (5,)
    for v in vectors.values():
        assert v.shape == (1, 192) and abs(float(v.norm()) - 1) < 1e-5
    assert 0.0 <= eer[0] <= 1.0 and dcf[0] >= 0.0
    # segments: 3 crops of 1 s per utterance -> [3, 192] per file, scores = mean over the 3 x 3 pairs
    tr3 = VerificationTrials(os.path.join(root, "trials.txt"), num_samples=16000, mode_eval=True, num_eval=3,
                             trials_dir=root)
    v3 = sugar_amd.extract_vectors(model, tr3.utterances, device="cuda:0", batch_size=2)
    assert all(v.shape == (3, 192) for v in v3.values())
    s3, labs, pairs = sugar_amd.score_trials(tr3, v3)
    want = [torch.nn.functional.cosine_similarity(v3[e].unsqueeze(-1), v3[t].unsqueeze(-1).transpose(0, 2)).mean().item()
            for e, t in pairs]
    assert np.abs(s3 - np.array(want)).max() < 1e-5


@pytest.mark.gpu
def test_native_file_feed_equals_python_loader(tmp_path):
    """An Utterance list of 16-bit PCM files goes through the host feed's native reader (sugar_wav_probe / sugar_wav_load:
    no per-item Python); any other dataset -- here a subclass with the same items -- through its own __getitem__, on loader
    threads.  Same files, same batches: the vectors are bit-equal, for whole utterances and for crops, with one packing
    thread and with several, and the feed says which reader it used."""
    root = str(tmp_path)
    write_files(root)

    class PyUtterance(Utterance):        # same items, but not the class the native reader recognises
        pass

    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    sub = sup.clone(layout.SMALLEST).set_precision("bf16")
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).cuda().eval()
    for num_samples, num_eval in ((0, 1), (16000, 3)):
        tr = VerificationTrials(os.path.join(root, "trials.txt"), num_samples=num_samples, mode_eval=True,
                                num_eval=num_eval, trials_dir=root)
        u = tr.utterances
        py = PyUtterance(u.datalst, u.num_samples, u.mode_eval, u.num_eval, u.root_dir)
        got = {}
        for name, data, jobs in (("native", u, 1), ("native4", u, 4), ("python", py, 1), ("python4", py, 4)):
            v = sugar_amd.extract_vectors(model, data, device="cuda:0", n_jobs=jobs, batch_size=3)
            assert bool(sugar_amd.extract_vectors.last_hostfeed_stats.get("native_reader")) == name.startswith("native")
            assert list(v.keys()) == [u.path(i) for i in range(len(u))]
            got[name] = v.matrix.clone()
        for name in ("native4", "python", "python4"):
            assert torch.equal(got[name], got["native"]), (num_samples, name)
        staged = sugar_amd.extract_vectors(model, u.stage("cuda:0", batch_size=3))
        assert torch.equal(staged.matrix, got["native"])


@pytest.mark.gpu
def test_streamed_feed_in_several_chunks_equals_one_chunk(tmp_path, monkeypatch):
    """The host feed sorts and batches the list chunk by chunk (2^20 frames at a time, so that the GPU starts before the
    list is read and host memory stays bounded).  With a chunk of 1 500 frames this 12-file list goes through in
    several chunks -- other batch compositions, same row order: the table equals the one-chunk result (bit for bit: an
    utterance's embedding does not depend on what it is batched with), for the native reader and for the Python loader."""
    import sugar_amd.vectors as V
    from scipy.io import wavfile
    root = str(tmp_path)
    rng = np.random.default_rng(11)
    names = []
    for i, n in enumerate(rng.integers(20000, 90000, size=12)):         # 1.25 .. 5.6 s (the short ones wrap-padded to 3 s)
        names.append("c%02d.wav" % i)
        wavfile.write(os.path.join(root, names[-1]), 16000, (rng.standard_normal(n) * 2000).astype(np.int16))
    u = Utterance(names, num_samples=0, mode_eval=True, num_eval=1, root_dir=root)

    class PyUtterance(Utterance):
        pass

    py = PyUtterance(u.datalst, u.num_samples, u.mode_eval, u.num_eval, u.root_dir)
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    sub = sup.clone(layout.SMALLEST).set_precision("bf16")
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).cuda().eval()
    want = sugar_amd.extract_vectors(model, u, device="cuda:0", batch_size=4)
    calls = []
    orig = sugar_amd.SpeakerModel._embed_flat
    monkeypatch.setattr(sugar_amd.SpeakerModel, "_embed_flat",
                        lambda self, s, flat, plan, nrm: (calls.append(plan.B), orig(self, s, flat, plan, nrm))[1])
    monkeypatch.setattr(V._HostFeed, "CHUNK_FRAMES", 1500)
    for data in (u, py):
        del calls[:]
        got = sugar_amd.extract_vectors(model, data, device="cuda:0", batch_size=4, n_jobs=2)
        assert sum(calls) == len(u) and len(calls) > -(-len(u) // 4)          # more, smaller batches than one chunk gives
        assert list(got.keys()) == list(want.keys())
        assert torch.equal(got.matrix, want.matrix)
