"""CPU: host-side mirror of the reference interface, checkpoint format, C-ABI surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import sugar_amd
from conftest import ROOT, cfg_of
from sugar_amd import _lib, layout, synth
from sugar_amd.vectors import plan_batches, shard_indices


def test_supernet_layout_is_the_reference_layout(meta, supernet):
    assert [[k, list(v.shape)] for k, v in supernet.state_dict().items()] == meta["supernet_keys"]
    assert sum(p.numel() for p in supernet.parameters()) == meta["supernet_numel"] == 7550818


def test_clone_reproduces_reference_clone(golden, meta, supernet):
    for name in meta["configs"]:
        sub = supernet.clone(cfg_of(meta, name))
        sd = sub.state_dict()
        assert [[k, list(v.shape)] for k, v in sd.items()] == meta["subnet_keys"][name]
        sums = np.array([float(v.double().sum()) for v in sd.values()])
        assert np.allclose(sums, golden["clone_sums/" + name], rtol=1e-6, atol=1e-5), name
    for key in golden.files:
        if key.startswith("clone_tensor/"):
            _, name, k = key.split("/", 2)
            sd = supernet.clone(cfg_of(meta, name)).state_dict()
            assert np.allclose(sd[k].numpy(), golden[key], rtol=1e-6, atol=1e-7), key
    supernet.config = None


def test_config_protocol(supernet):
    supernet.config = None
    assert supernet.config == (4, [512] * 5, [5] * 5, 1536)
    supernet.config = layout.BASE
    assert supernet.config[0] == 3 and supernet.active_catconv_channels == 1536
    with pytest.raises(ValueError):
        supernet.config = (5, [512] * 6, [5] * 6, 1536)          # deeper than the supernet
    with pytest.raises(ValueError):
        supernet.config = (3, [1024] * 4, [5] * 4, 1536)         # wider than the supernet
    with pytest.raises(ValueError):
        supernet.config = (3, [512] * 4, [7, 3, 3, 3], 1536)     # kernel not in the supernet
    supernet.config = None


def test_count_ops_params_known_answers(meta, supernet):
    for key, (ops, params) in meta["count_ops_params"].items():
        name, T = key.split("@")
        supernet.config = cfg_of(meta, name)
        assert supernet.count_ops([1, 80, int(T)]) == ops
        assert supernet.count_params() == params
    supernet.config = None


def test_tdnnpath_sampler_stream(meta, supernet):
    space = supernet.space(mode=["kernel", "depth", "width1", "width2"], verbose=False)
    assert space == meta["tdnnpath_space"]
    path = sugar_amd.TDNNPath(space)
    got = [path.sample() for _ in range(16)]
    assert [[d, list(c), list(k), cc] for d, c, k, cc in got] == meta["tdnnpath_samples"]


def test_config_names():
    name = "ecapa-tdnn.3.512.512.512.512.5.3.3.3.1536"
    assert layout.config_from_name("depth." + name + ".bn.tar") == layout.BASE
    assert layout.config_name(layout.BASE) == name


def test_from_pretrained_checkpoint_format(tmp_path, sup_state, meta):
    # supernet file: {'state_dict': {'module.__S__.*', 'module.__L__.W', 'module.__T__.*'}}; subnet file: {'subnet', 'bn', ...}
    ck = {"module.__S__." + k: v for k, v in sup_state.items()}
    ck["module.__L__.W"] = torch.zeros(192, 10)
    ck["module.__T__.torchfb.0.weight"] = torch.tensor([[[-0.97, 1.0]]])
    torch.save({"state_dict": ck, "epoch": 1}, tmp_path / "depth.torchparams")
    shapes = layout.subnet_shapes(layout.BASE)
    bn = synth.fill_state_dict({k: shapes[k] for k in meta["bn_file_keys"]}, seed=11)
    torch.save({"subnet": layout.BASE, "bn": bn, "EER (%) w/o AS-Norm": "1.14"}, tmp_path / "base.bn.tar")
    subnet, info = sugar_amd.WrappedModel.from_pretrained(str(tmp_path), "depth.torchparams", "base.bn.tar")
    assert isinstance(subnet.module.__S__, sugar_amd.TDNNSubnet)
    assert isinstance(subnet.module.__T__, sugar_amd.LogMelFbanks)
    sd = subnet.module.__S__.state_dict()
    assert torch.equal(sd["layer1.bn.running_mean"], bn["layer1.bn.running_mean"])        # BN file overrides the clone
    assert torch.equal(sd["layer1.conv.weight"], sup_state["layer1.conv.conv.weight"])      # K0 = 5: no transform
    assert info["subnet"]["subnet"] == layout.BASE
    assert set(sugar_amd.bn_state_dict(subnet.module.__S__).keys()) == set(meta["bn_file_keys"])
    with pytest.raises(ValueError):
        sugar_amd.WrappedModel.from_pretrained(str(tmp_path), "depth.torchparams", "base.bn.tar", feat_type="mfcc")
    with pytest.raises(ValueError):
        sugar_amd.WrappedModel.from_pretrained(str(tmp_path), "depth.torchparams", "base.bn.tar", model_type="resnet")
    with pytest.raises(FileNotFoundError):
        sugar_amd.WrappedModel.from_pretrained(str(tmp_path), "missing.torchparams", "base.bn.tar")


def test_no_cpu_fallback(supernet):
    sub = supernet.clone(layout.SMALLEST)
    with pytest.raises(_lib.SugarError):
        sub(torch.zeros(1, 80, 63))
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks()))
    with pytest.raises(_lib.SugarError):
        model(torch.zeros(1, 10000))
    with pytest.raises(_lib.SugarError):
        sugar_amd.scores.score_pairs(torch.zeros(2, 192), torch.zeros(3, dtype=torch.int32),
                                     torch.zeros(1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32))
    supernet.config = None


def test_cabi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "sugar_amd.h")).read()
    declared = set(re.findall(r"\b(sugar_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert b"gfx950" in _lib.lib().sugar_version()
    # argument checking that needs no GPU: NULL pointers are rejected with SUGAR_EINVAL and a message
    rc = _lib.lib().sugar_cosine_trials(None, 192, None, 1, None, None, 1, None, None)
    assert rc == -1 and b"null" in _lib.lib().sugar_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "sugar_cosine_trials")


def test_struct_layout_matches_header():
    # SugarBlock: 2 ints + first(1+4) + res(7 + 28) + last(1+4) + se(4) pointers
    assert ctypes.sizeof(_lib.SugarBN) == 32
    assert ctypes.sizeof(_lib.SugarBlock) == 8 + 8 * (5 + 35 + 5 + 4)
    assert ctypes.sizeof(_lib.SugarSubnetDesc) == 32 + 8 * 5 + 8 * ctypes.sizeof(_lib.SugarBlock) + 8 * 3 + 32 + 16 + 32


def test_batch_planning():
    lengths = [32000, 320000, 48000, 48000, 100000, 64000]
    batches = plan_batches(lengths, batch_size=4, max_frames=1200)
    flat = sorted(i for b in batches for i in b)
    assert flat == list(range(len(lengths)))
    for b in batches:
        assert len(b) <= 4
        assert len(b) == 1 or sum(1 + lengths[i] // 160 for i in b) <= 1200
    parts = [shard_indices(10, r, 4) for r in range(4)]
    assert sorted(i for p in parts for i in p) == list(range(10))
    w = [5, 1, 9, 3, 7, 2, 8]
    parts = [shard_indices(7, r, 2, w) for r in range(2)]
    assert sorted(i for p in parts for i in p) == list(range(7))
    loads = [sum(w[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= max(w)


def test_batchplan_rejects_short_audio():
    from sugar_amd.transforms import BatchPlan, num_frames
    assert num_frames(48000) == 301 and num_frames(10000) == 63
    with pytest.raises(ValueError):
        BatchPlan([200], "cpu")
    p = BatchPlan([10000, 48000], "cpu")
    assert p.R == 364 and p.t_max == 301 and p.row_off.tolist() == [0, 63, 364] and p.wav_off.tolist() == [0, 10000]


def test_packed_fp32_forms_are_confined_to_the_probe_cleared_ones():
    """One packed-FP32 operand form (op_sel bit of src1 set: the low result takes the high half of src1) gives wrong
    results next to MFMA waves of another kernel on MI355X.  The library may use packed FP32,
    but only in forms tools/pkfp32_probe.hip ran clean beside an MFMA burner; the build reads every source's ISA."""
    import os
    from sugar_amd import build
    for form in build.PACKED_FP32_SAFE:
        sel = [t for t in form.split() if t.startswith("op_sel:")]
        assert not sel or sel[0].split(",")[1].strip("[]") == "0", form     # never the high half of src1 into the low result
    assert "op_sel:[0,1] op_sel_hi:[1,0]" not in build.PACKED_FP32_SAFE
    if os.environ.get("SUGAR_PACKED_FP32_ALL") is None:
        assert build.PACKED_FP32_FILES == {"fbank.hip"}
        for src in build.SOURCES:
            assert ("-packed-fp32-ops" in build.EXTRA_FLAGS.get(src, [])) == (src != "fbank.hip"), src
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if os.path.exists(hipcc):
        assert build._check_packed_forms(hipcc, "fbank.hip") > 0


def test_pack_segments_host_only():
    """sugar_pack_segments (host memory, no GPU call): ragged float32 segments packed back to back by 1 .. 64 threads over
    equal byte ranges -- boundaries fall inside segments -- equal torch.cat; null / negative arguments are refused."""
    import ctypes
    segs = [torch.randn(n) for n in (5, 70000, 1, 333333, 99999, 257)]
    n, want = len(segs), torch.cat(segs)
    src = (ctypes.c_void_p * n)(*[s.data_ptr() for s in segs])
    lens = (ctypes.c_int64 * n)(*[s.numel() for s in segs])
    for nt in (1, 3, 8, 64, 0):
        dst = torch.full((want.numel() + 8,), -7.0)
        assert _lib.lib().sugar_pack_segments(src, lens, n, dst.data_ptr(), nt) == 0
        assert torch.equal(dst[:-8], want) and (dst[-8:] == -7.0).all(), nt
    assert _lib.lib().sugar_pack_segments(None, lens, n, dst.data_ptr(), 4) < 0
    bad = (ctypes.c_int64 * n)(*([-1] + [1] * (n - 1)))
    assert _lib.lib().sugar_pack_segments(src, bad, n, dst.data_ptr(), 4) < 0


def test_vector_table_is_a_lazy_dict_with_dict_semantics():
    """extract_vectors' return value: views made on first use (4 874 views up front were 22 ms of host time per call),
    same contents and key order as the dict the reference builds (sugar/vectors.py:148-153), a repeated file keeping its
    LAST rows."""
    import pickle
    from sugar_amd.vectors import VectorTable
    m = torch.randn(10, 4)
    files, counts = ["a", "b", "c", "a"], [2, 3, 1, 4]
    ref = {}
    start = 0
    for f, n in zip(files, counts):
        ref[f] = m[start:start + n]
        start += n
    t = VectorTable(files, counts, m)
    assert dict.__len__(t) == 0                                # nothing materialised yet
    assert len(t) == 3 and list(t) == list(ref) and list(t.keys()) == list(ref) and "b" in t and "z" not in t
    assert torch.equal(t["a"], ref["a"]) and torch.equal(t.get("c"), ref["c"]) and t.get("z") is None
    assert dict.__len__(t) == 2                                # only what was asked for
    with pytest.raises(KeyError):
        t["zz"]
    assert t.seg_off.tolist() == [0, 2, 5, 6, 10]
    for k, v in VectorTable(files, counts, m).items():
        assert torch.equal(v, ref[k])
    assert [k for k in dict(VectorTable(files, counts, m))] == list(ref)
    u = VectorTable(files, counts, m)
    u["q"] = m[:1]                                             # mutation: an ordinary dict from here on
    assert len(u) == 4 and torch.equal(u["a"], ref["a"]) and list(u)[-1] == "q"
    p = pickle.loads(pickle.dumps(VectorTable(files, counts, m)))
    assert type(p) is dict and torch.equal(p["b"], ref["b"])
    e = VectorTable([], [], torch.empty(0, 4))
    assert len(e) == 0 and list(e.items()) == [] and not e


def test_native_wav_reader_equals_loadWAV(tmp_path):
    """sugar_wav_probe / sugar_wav_load (the host feed's reader for evaluation lists of 16-bit PCM files) against
    loadWAV in eval mode, sample for sample: short files (wrap padding to 48 001), exactly 48 000, long ones, whole
    utterances and linspace crops, a file with a LIST chunk in front of its data and one with an odd-sized chunk; other
    formats are reported as such (the caller then keeps the Python loader)."""
    import ctypes
    import struct
    from scipy.io import wavfile
    from sugar_amd.waveforms import eval_segments, loadWAV
    rng = np.random.default_rng(5)
    sizes = [700, 16000, 47999, 48000, 48001, 50000, 123457, 320000]
    paths = []
    for i, n in enumerate(sizes):
        p = str(tmp_path / ("u%d.wav" % i))
        wavfile.write(p, 16000, rng.integers(-32768, 32768, size=n).astype(np.int16))
        paths.append(p)
    # a LIST chunk (odd size: padded to even) between fmt and data, as ffmpeg / sox write them
    raw = open(paths[5], "rb").read()
    k = raw.index(b"data")
    extra = b"LIST" + struct.pack("<I", 7) + b"INFOabc" + b"\x00"
    body = raw[:k] + extra + raw[k:]
    body = body[:4] + struct.pack("<I", len(body) - 8) + body[8:]
    p = str(tmp_path / "with_list.wav")
    open(p, "wb").write(body)
    paths.append(p)
    sizes.append(sizes[5])
    assert (wavfile.read(p)[1] == wavfile.read(paths[5])[1]).all()
    n = len(paths)
    lib = _lib.lib()
    arr = (ctypes.c_char_p * n)(*[os.fsencode(q) for q in paths])
    size, off = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
    _lib.check(lib.sugar_wav_probe(arr, n, 16000, size.ctypes.data, off.ctypes.data, 3), "sugar_wav_probe")
    assert size.tolist() == sizes
    for num_samples, num_eval in ((0, 1), (48000, 6), (32000, 10), (16000, 1)):
        starts, lengths = eval_segments(size, num_samples, num_eval)
        n_seg = starts.shape[1]
        f_idx = np.repeat(np.arange(n, dtype=np.int32), n_seg)
        s_len = np.repeat(lengths, n_seg)
        s_start = np.ascontiguousarray(starts.reshape(-1))
        for threads in (1, 5):
            dst = np.full(int(s_len.sum()), np.nan, dtype=np.float32)
            _lib.check(lib.sugar_wav_load(arr, size.ctypes.data, off.ctypes.data, n, f_idx.ctypes.data, s_start.ctypes.data,
                                          s_len.ctypes.data, len(f_idx), dst.ctypes.data, threads), "sugar_wav_load")
            want = np.concatenate([loadWAV(q, num_samples, True, num_eval).reshape(-1) for q in paths])
            assert dst.shape == want.shape and (dst == want).all(), (num_samples, num_eval, threads)
    bad = np.array([n], dtype=np.int32)                         # a segment naming a file past the list: refused
    one = np.array([10], dtype=np.int64)
    assert lib.sugar_wav_load(arr, size.ctypes.data, off.ctypes.data, n, bad.ctypes.data, one.ctypes.data, one.ctypes.data,
                              1, dst.ctypes.data, 1) != 0
    # other formats: stereo, 8 kHz, float32 -> -1 (not an error); a missing file -> an error
    others = []
    for name, rate, data in (("stereo", 16000, rng.integers(-9, 9, size=(4000, 2)).astype(np.int16)),
                             ("8k", 8000, rng.integers(-9, 9, size=4000).astype(np.int16)),
                             ("f32", 16000, rng.standard_normal(4000).astype(np.float32))):
        q = str(tmp_path / (name + ".wav"))
        wavfile.write(q, rate, data)
        others.append(q)
    arr2 = (ctypes.c_char_p * 3)(*[os.fsencode(q) for q in others])
    s2, o2 = np.empty(3, dtype=np.int64), np.empty(3, dtype=np.int64)
    _lib.check(lib.sugar_wav_probe(arr2, 3, 16000, s2.ctypes.data, o2.ctypes.data, 2), "sugar_wav_probe")
    assert s2.tolist() == [-1, -1, -1]
    arr3 = (ctypes.c_char_p * 1)(os.fsencode(str(tmp_path / "missing.wav")))
    assert lib.sugar_wav_probe(arr3, 1, 16000, s2.ctypes.data, o2.ctypes.data, 1) != 0


def test_dataset_walk_keeps_order_and_surfaces_loader_errors():
    """vectors._walk: items in list order whether they are loaded in line (cheap items: an in-memory list) or by loader
    threads running ahead (slow items: files); an exception in a loader thread reaches the consumer; abandoning the
    walk half way does not hang."""
    import time
    from sugar_amd.vectors import _walk

    class Data:
        def __init__(self, n, delay, bad=None):
            self.n, self.delay, self.bad, self.seen = n, delay, bad, []

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            if self.delay:
                time.sleep(self.delay)
            if i == self.bad:
                raise OSError("cannot read item %d" % i)
            self.seen.append(i)
            return {"file": "f%d" % i, "input": torch.zeros(1, 300)}

    for delay, jobs in ((0, 8), (0.002, 8), (0.002, 1), (0, 1)):
        d = Data(150, delay)
        t0 = time.perf_counter()
        assert [it["file"] for it in _walk(d, jobs)] == ["f%d" % i for i in range(150)]
        if delay and jobs > 1:
            assert time.perf_counter() - t0 < 150 * delay * 0.6        # really in parallel
    with pytest.raises(OSError, match="item 97"):
        list(_walk(Data(150, 0.002, bad=97), 8))
    d = Data(400, 0.002)
    for k, it in enumerate(_walk(d, 8)):
        if k == 20:
            break                                                     # the generator is closed: its pool shuts down
    assert len(d.seen) < 200
    assert list(_walk(Data(0, 0), 4)) == [] and len(list(_walk(Data(3, 0), 4))) == 3
