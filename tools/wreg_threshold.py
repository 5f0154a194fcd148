"""In-pipeline A/B of the weights-in-registers GEMM against the tile kernel over batch sizes (where should SUGAR_GEMM_WREG_MIN_R sit?)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import sugar_amd
from sugar_amd import layout, synth
dev = torch.device("cuda", 0)
sup = sugar_amd.tdnn8m2g(); sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(), seed=0))
sub = sup.clone(layout.BASE).set_precision("bf16")
model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).eval()
for B in (96, 128, 160, 192, 224, 256, 384):
    wav = torch.randn(B, 48000, device=dev)
    res = []
    for rep in range(2):
        for mode in ("0", str(1 << 40)):
            os.environ["SUGAR_GEMM_WREG_MIN_R"] = mode
            for _ in range(5): model(wav, normalize=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): model(wav, normalize=True)
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 30 * 1e3)
    print("B=%d R=%d  wreg %.4f / %.4f ms   tiles %.4f / %.4f ms" % (B, B * 301, res[0], res[2], res[1], res[3]), flush=True)
