"""
GPU: the allocation-free front-end entry point (sugar_fbank_fwd_noalloc, include/sugar_amd.h) against the existing
sugar_fbank_fwd, both through the C-ABI.  Same kernels, so the bar is bit-exact; two calls in a row, the second with a
smaller input (the static scratch keeps the first call's contents), f32 and bf16 outputs, ragged utterances, one
utterance long enough for the chunked instance norm (> 320 frames), then the size limit.
Collected last on purpose (the round-end run is `pytest -x`).
"""
import pytest
import torch

from sugar_amd import _lib, synth
from sugar_amd.transforms import BatchPlan

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(lens, seed, tdt, noalloc):
    wavs = synth.ragged_waveforms(lens, seed=seed)
    flat = torch.cat(wavs).to(DEV)
    plan = BatchPlan(lens, DEV)
    feats = torch.full((plan.R, 80), float("nan"), dtype=tdt, device=DEV)
    dt = _lib.SUGAR_F32 if tdt == torch.float32 else _lib.SUGAR_BF16
    stream = torch.cuda.current_stream().cuda_stream
    if noalloc:
        rc = _lib.lib().sugar_fbank_fwd_noalloc(flat.data_ptr(), plan.wav_off.data_ptr(), plan.wav_len.data_ptr(),
                                                plan.row_off.data_ptr(), plan.B, plan.R, plan.t_max, feats.data_ptr(),
                                                dt, stream)
    else:
        nbytes = _lib.lib().sugar_fbank_scratch_bytes(plan.B, plan.R)
        logmel = torch.empty(nbytes // 4, dtype=torch.float32, device=DEV)
        rc = _lib.lib().sugar_fbank_fwd(flat.data_ptr(), plan.wav_off.data_ptr(), plan.wav_len.data_ptr(),
                                        plan.row_off.data_ptr(), plan.B, plan.R, plan.t_max, logmel.data_ptr(),
                                        feats.data_ptr(), dt, stream)
    _lib.check(rc, "fbank")
    torch.cuda.synchronize()
    return feats.cpu()


@pytest.mark.parametrize("tdt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_noalloc_fbank_equals_fbank_twice_second_smaller(tdt):
    big = [48000] * 6 + [66666, 257, 10240, 160000]        # 160000 samples = 1001 frames: chunked instance norm
    small = [10000, 32000, 4321]
    for lens, seed in ((big, 11), (small, 12)):            # two calls in a row on the same static scratch
        got = _run(lens, seed, tdt, noalloc=True)
        want = _run(lens, seed, tdt, noalloc=False)
        assert torch.isfinite(got.float()).all()
        assert torch.equal(got, want), (lens, (got.float() - want.float()).abs().max())


def test_noalloc_fbank_rejects_oversize_and_takes_the_limit():
    lib = _lib.lib()
    dummy = torch.zeros(16, device=DEV)
    for B, R in ((257, 1000), (4, 16385)):
        rc = lib.sugar_fbank_fwd_noalloc(dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(), B, R, 10,
                                         dummy.data_ptr(), _lib.SUGAR_F32, None)
        assert rc == -1 and b"static scratch" in lib.sugar_last_error()
    # exactly at the frame limit: 16384 frames = 64 utterances of 256 frames (40 800 samples -> 1 + 255 frames)
    lens = [40800 + 159] * 64
    assert BatchPlan(lens, DEV).R == 16384
    assert torch.equal(_run(lens, 13, torch.float32, True), _run(lens, 13, torch.float32, False))


def test_noalloc_fbank_runs_under_stream_capture_and_allocates_nothing():
    """The entry point's defining property: after its one-off set-up it allocates nothing and never synchronises, so it
    can be captured into a HIP graph (a hipMalloc or a sync inside a capturing stream fails the capture) and the device's
    free memory does not move over repeated calls."""
    lens = [48000] * 4 + [20000, 7777]
    wavs = synth.ragged_waveforms(lens, seed=21)
    flat = torch.cat(wavs).to(DEV)
    plan = BatchPlan(lens, DEV)
    want = _run(lens, 21, torch.bfloat16, noalloc=False)
    feats = torch.zeros((plan.R, 80), dtype=torch.bfloat16, device=DEV)
    lib = _lib.lib()

    def call(stream):
        _lib.check(lib.sugar_fbank_fwd_noalloc(flat.data_ptr(), plan.wav_off.data_ptr(), plan.wav_len.data_ptr(),
                                               plan.row_off.data_ptr(), plan.B, plan.R, plan.t_max, feats.data_ptr(),
                                               _lib.SUGAR_BF16, stream), "fbank_noalloc")

    call(torch.cuda.current_stream().cuda_stream)             # warm call: per-device set-up (hipGetSymbolAddress) done
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(8):
        call(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0               # hipMemGetInfo: no device allocation in 8 calls
    feats.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert not feats.any()                                     # captured, not run
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(feats.cpu(), want)
