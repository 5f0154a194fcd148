"""CPU: the oracle (oracle/ref_cpu.py) against the golden vectors produced by the real reference (oracle/make_golden.py)."""
import numpy as np
import torch

from conftest import cfg_of
from oracle import ref_cpu
from sugar_amd import layout, synth


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_preemphasis_golden(golden):
    w = synth.waveforms(2, 64, seed=3)
    assert np.array_equal(ref_cpu.preemphasis(w).numpy(), golden["preemph/out"])


def test_mel_filterbank_known_answers(meta):
    fb = ref_cpu.mel_filterbank()
    assert fb.shape == (257, 80)
    assert int((fb > 0).sum()) == meta["mel_nnz"] == 477
    assert meta["mel_empty_bins"] == 0 and int(((fb > 0).sum(0) == 0).sum()) == 0
    assert float(fb[0].abs().max()) == 0.0 and float(fb[256].abs().max()) == 0.0   # DC and Nyquist carry no weight


def test_fbank_two_restatements_agree(golden):
    """PARITY UNPINNED stage (torchaudio absent): torch.stft restatement vs independent float64 numpy restatement."""
    wav = synth.waveforms(1, 10000, seed=0)
    fb = ref_cpu.logmel_fbank(wav)
    assert fb.shape == (1, 1, 80, 63)
    assert np.abs(fb[0, 0].numpy() - golden["fbank/cfg1"]).max() < 1e-5
    assert np.abs(fb[0, 0].numpy() - ref_cpu.logmel_fbank_np64(wav[0].numpy())).max() < 2e-4
    assert ref_cpu.logmel_fbank(synth.waveforms(2, 48000, seed=1)).shape == (2, 1, 80, 301)


def test_clone_state_matches_reference_clone(golden, meta, sup_state):
    for name in meta["configs"]:
        cfg = cfg_of(meta, name)
        sd = ref_cpu.clone_state(sup_state, cfg)
        assert [[k, list(v.shape)] for k, v in sd.items()] == meta["subnet_keys"][name]
        sums = np.array([float(v.double().sum()) for v in sd.values()])
        asums = np.array([float(v.double().abs().sum()) for v in sd.values()])
        assert np.allclose(sums, golden["clone_sums/" + name], rtol=1e-6, atol=1e-5)
        assert np.allclose(asums, golden["clone_abssums/" + name], rtol=1e-6, atol=1e-5)
    for key in golden.files:
        if key.startswith("clone_tensor/"):
            _, name, k = key.split("/", 2)
            sd = ref_cpu.clone_state(sup_state, cfg_of(meta, name))
            assert np.allclose(sd[k].numpy(), golden[key], rtol=1e-6, atol=1e-7), key


def test_backbone_stages_match_reference(golden, meta, sup_state):
    x = ref_cpu.logmel_fbank(synth.waveforms(1, 10000, seed=0)).squeeze(1)
    for name in ("smallest", "odd"):
        cfg = cfg_of(meta, name)
        stages = {}
        ref_cpu.subnet_forward(ref_cpu.clone_state(sup_state, cfg), cfg[0], x, stages)
        for k, v in stages.items():
            assert _rel(v[0].numpy(), golden["stage/%s/%s" % (name, k)]) < 2e-5, (name, k)


def test_embeddings_match_reference(golden, meta, sup_state):
    x63 = ref_cpu.logmel_fbank(synth.waveforms(1, 10000, seed=0)).squeeze(1)
    base = ref_cpu.clone_state(sup_state, layout.BASE)
    assert _rel(ref_cpu.subnet_forward(base, 3, x63).numpy(), golden["embed/base/cfg1"]) < 1e-5
    x301 = ref_cpu.logmel_fbank(synth.waveforms(4, 48000, seed=0)).squeeze(1)
    for name in ("base", "largest", "rand0", "rand1", "rand2", "odd"):
        cfg = cfg_of(meta, name)
        e = ref_cpu.subnet_forward(ref_cpu.clone_state(sup_state, cfg), cfg[0], x301)
        assert _rel(e.numpy(), golden["embed/%s/4x3s" % name]) < 2e-5, name


def test_ragged_one_at_a_time(golden, meta, sup_state):
    wavs = synth.ragged_waveforms(meta["ragged_lengths"], seed=7)
    e = ref_cpu.embed_ragged(ref_cpu.clone_state(sup_state, layout.BASE), 3, wavs)
    assert _rel(e.numpy(), golden["embed/base/ragged"]) < 2e-5


def test_long_utterances_match_reference(sup_state):
    """Whole-utterance evaluation of 7-21 s audio (tests/golden/long.npz, made by oracle/make_golden_long.py from the
    reference's own cloned subnets): the lengths at which the GPU path cuts utterances into time segments."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "long.npz"))
    wavs = synth.ragged_waveforms([int(n) for n in g["lengths"]], seed=int(g["seed"]))
    osd = ref_cpu.clone_state(sup_state, layout.BASE)
    got = ref_cpu.embed_ragged(osd, layout.BASE[0], wavs)
    want = torch.from_numpy(g["embed/base/long"])
    assert float((got - want).abs().max() / want.abs().max()) < 2e-5


def test_bn_file_override(golden, meta, sup_state):
    sd = ref_cpu.clone_state(sup_state, layout.BASE)
    shapes = layout.subnet_shapes(layout.BASE)
    sd.update(synth.fill_state_dict({k: shapes[k] for k in meta["bn_file_keys"]}, seed=11))
    x63 = ref_cpu.logmel_fbank(synth.waveforms(1, 10000, seed=0)).squeeze(1)
    assert _rel(ref_cpu.subnet_forward(sd, 3, x63).numpy(), golden["embed/base/cfg1_bnfile"]) < 1e-5


def test_cosine_scoring_matches_calculate_score(golden, meta):
    segs = meta["score_segs"]
    vecs = [synth.normal("score/vec%d" % i, (segs[i], 192), seed=2) for i in range(len(segs))]
    vecs = [torch.nn.functional.normalize(v, p=2, dim=1) if i % 2 == 0 else v for i, v in enumerate(vecs)]
    _, a, b = synth.trial_pairs(len(segs), 300, seed=2)
    assert np.array_equal(ref_cpu.score_trials(a, b, vecs), golden["score/scores"])


def test_mac_known_answers():
    # SURVEY 8d / BASELINE.md section 2: conv + linear MACs per utterance
    assert abs(ref_cpu.count_macs(layout.BASE, 301) / 1.4423e9 - 1) < 1e-3
    assert abs(ref_cpu.count_macs(layout.LARGEST, 301) / 1.9319e9 - 1) < 1e-3


def test_oracle_norm2_scoring_matches_reference_outputs():
    """oracle norm2_score / list-of-modes scoring against the reference's own calculate_score / score_trials outputs
    (tests/golden/norm2.npz, generated by oracle/make_golden_norm2.py from sugar/scores.py:114-216)."""
    import os
    import numpy as np
    from oracle import ref_cpu
    from oracle.make_golden_norm2_inputs import inputs
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "norm2.npz"))
    vecs, lab, a, b = inputs()
    assert np.abs(ref_cpu.score_trials(a, b, vecs, mode="norm2") - gold["norm2"]).max() < 1e-6
    assert np.abs(ref_cpu.score_trials(a, b, vecs, mode=["cosine", "norm2"]) - gold["cosine_norm2"]).max() < 1e-6
    assert abs(ref_cpu.norm2_score(vecs[1], vecs[2]) - float(gold["default_1_2"])) < 1e-6
