"""
Host utilities next to the path -- calculate_val, wav_segment, txt2dict, bn_state_dict / load_bn_state_dict, keep_params
(sugar/metrics.py:221-281, sugar/waveforms.py:118-146, sugar/database.py:39-69, sugar/utils/utility.py:9-42, 67-75) --
against outputs of the reference's own functions (tests/golden/utils.npz, oracle/make_golden_utils.py).
"""
import os

import numpy as np
import pytest
import torch

import sugar_amd
from oracle.utils_inputs import val_inputs, wav_inputs
from sugar_amd import layout, synth
from sugar_amd.database import txt2dict
from sugar_amd.waveforms import wav_segment

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "utils.npz"))


def test_calculate_val_matches_reference():
    y, s1, s2 = val_inputs()
    for far in (1e-1, 1e-2):
        assert np.array_equal(np.array(sugar_amd.calculate_val(y, s1, far=far), dtype=np.float64), GOLD["val_%g" % far])
        got = sugar_amd.calculate_val(y, np.stack([s1, s2], axis=1), far=far)
        assert np.array_equal(np.array(got, dtype=np.float64), GOLD["val2d_%g" % far])


def test_wav_segment_and_txt2dict(tmp_path):
    for name, (audio, n, nov) in wav_inputs().items():
        assert np.array_equal(wav_segment(audio, n, nov), GOLD["wavseg_" + name]), name
    p = tmp_path / "utts.txt"
    p.write_text("id01 a/1.wav\nid02 b/1.wav\nid01 a/2.wav\n")
    assert txt2dict(str(p), "/data") == {"id01": ["/data/a/1.wav", "/data/a/2.wav"], "id02": ["/data/b/1.wav"]}


def test_bn_state_dict_round_trip_and_keep_params(tmp_path):
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
    sub = sup.clone(layout.SMALLEST)
    bn = sugar_amd.bn_state_dict(sub)
    assert "\n".join(bn.keys()) == str(GOLD["bn_keys"])                       # the reference's key list, in its order
    assert abs(sum(float(v.double().sum()) for v in bn.values()) - float(GOLD["bn_sum"])) < 1e-6
    sub2 = sup.clone(layout.SMALLEST)
    zeros = {k: torch.zeros_like(v) for k, v in bn.items()}
    sugar_amd.load_bn_state_dict(sub2, zeros)
    assert all(not v.any() for v in sugar_amd.bn_state_dict(sub2).values())
    sugar_amd.load_bn_state_dict(sub2, bn)
    assert all(torch.equal(a, b) for a, b in zip(sugar_amd.bn_state_dict(sub2).values(), bn.values()))
    with pytest.raises(AssertionError):                                      # a name the model does not have
        sugar_amd.load_bn_state_dict(sub2, {"layer1.bn.nope": torch.zeros(1)})
    ck = str(tmp_path / "latest.pth.tar")
    torch.save({"epoch": 3, "train_step": 7, "arch": "x", "state_dict": {"w": torch.ones(2)}, "optimizer": {"o": 1},
                "scheduler": {"s": 2}, "lr_step": "epoch"}, ck)
    out = str(tmp_path / "p.torchparams")
    sugar_amd.keep_params(ck, out)
    assert sorted(torch.load(out, map_location="cpu", weights_only=True)) == ["arch", "epoch", "state_dict", "train_step"]
    with pytest.raises(AssertionError):
        sugar_amd.keep_params(ck, out)                                       # refuses to overwrite
