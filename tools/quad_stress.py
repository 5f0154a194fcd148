"""Race screen for the quadrant schedule of the persistent GEMM: catconv-shaped convs on fresh random inputs, N times,
each compared bit for bit with one workgroup per tile (gemm_no_persist) -- every second run while another stream keeps
the memory system busy.  python tools/quad_stress.py [runs]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sugar_amd import _lib

DEV = "cuda:0"


def conv(w, bn, x, rowd, B, tmax):
    cout, cin = w.shape[0], w.shape[1]
    R = x.shape[0]
    y = torch.full((R, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    cbn = _lib.SugarBN(*[_lib.fptr(t) for t in bn])
    rc = _lib.lib().sugar_conv1d_fwd(_lib.fptr(w), C.byref(cbn), cout, cin, 1, 1, 3, x.data_ptr(), None, y.data_ptr(),
                                     rowd.data_ptr(), B, R, tmax, _lib.SUGAR_BF16, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "sugar_conv1d_fwd")
    torch.cuda.synchronize()
    return y


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    bad = total = 0
    side = torch.cuda.Stream(DEV)
    junk = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    g = torch.Generator().manual_seed(1)
    for i in range(runs):
        for cin, cout, frames in ((1536, 1536, [301] * 110 + [70]), (1152, 1152, [301] * 100), (1024, 1024, [301] * 176)):
            R = sum(frames)
            off = [0]
            for t in frames:
                off.append(off[-1] + t)
            rowd = torch.tensor(off, dtype=torch.int32, device=DEV)
            x = torch.randn(R, cin, generator=g).to(torch.bfloat16).to(DEV)
            w = (torch.rand(cout, cin, 1, generator=g) * 2 - 1) * (3.0 / cin) ** 0.5
            bn = [torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
                  torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5]
            if i & 1:
                with torch.cuda.stream(side):
                    for _ in range(6):
                        junk.add_(1)
            got = conv(w, bn, x, rowd, len(frames), max(frames))
            with _lib.tuned(gemm_no_persist=1):
                want = conv(w, bn, x, rowd, len(frames), max(frames))
            total += 1
            if not torch.equal(got, want) or not torch.isfinite(got.float()).all():
                bad += 1
                print("run", i, (cin, cout), "DIFFERS:", int((got != want).sum()), "values", flush=True)
        print("run %d done, %d of %d differ so far" % (i, bad, total), flush=True)
    print("quadrant schedule vs one workgroup per tile: %d of %d comparisons differ" % (bad, total), flush=True)


if __name__ == "__main__":
    main()
