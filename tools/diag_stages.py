"""Diagnostic (GPU box): per-stage error of the HIP path vs the oracle for one config, both precisions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import sugar_amd
from oracle import ref_cpu
from sugar_amd import layout, synth

name = sys.argv[1] if len(sys.argv) > 1 else "largest"
cfg = {"largest": layout.LARGEST, "base": layout.BASE, "smallest": layout.SMALLEST}[name]
sd = synth.fill_state_dict(layout.supernet_shapes(), seed=0)
sup = sugar_amd.tdnn8m2g(); sup.load_state_dict(sd)
wav = synth.waveforms(2, 48000, seed=0)
x = ref_cpu.logmel_fbank(wav).squeeze(1)
osd = ref_cpu.clone_state(sd, cfg)
stages = {}
want = ref_cpu.subnet_forward(osd, cfg[0], x, stages)
B, _, T = x.shape
for prec in ("fp32", "bf16"):
    sub = sup.clone(cfg).set_precision(prec)
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    feats = x.transpose(1, 2).contiguous().view(B * T, 80).to(dt).cuda()
    row_off = torch.arange(0, (B + 1) * T, T, dtype=torch.int32).cuda()
    dims = sub.packed_dims()
    depth = cfg[0]
    taps = [(0, "layer1")]
    for i in range(depth):
        taps += [(1 + 4 * i, "b%d.first" % i), (2 + 4 * i, "b%d.res2" % i), (3 + 4 * i, "b%d.last" % i), (4 + 4 * i, "b%d.out" % i)]
    taps += [(100, "catconv")]
    print("==", name, prec)
    for what, key in taps:
        act, _ = sub.tap(feats, row_off, B, B * T, T, what)
        w = stages[key].transpose(1, 2).reshape(B * T, -1)
        a = act.cpu()[:, :w.shape[1]]
        rel = ((a - w).norm() / w.norm()).item()
        print("%-10s relL2 %.3e  max|ref| %.3e  mean|ref| %.3e" % (key, rel, w.abs().max().item(), w.abs().mean().item()))
    pooled, emb = sub.tap(feats, row_off, B, B * T, T, 101)
    ccat, cp = cfg[3], dims["ccatp"]
    p = pooled.cpu()
    got = torch.cat([p[:, :ccat], p[:, cp:cp + ccat]], 1)
    w = stages["pool"]
    print("pool mean  relL2 %.3e" % ((got[:, :ccat] - w[:, :ccat]).norm() / w[:, :ccat].norm()).item())
    print("pool std   relL2 %.3e" % ((got[:, ccat:] - w[:, ccat:]).norm() / w[:, ccat:].norm()).item())
    print("embed      relL2 %.3e cos %s" % (((emb.cpu() - want).norm() / want.norm()).item(),
                                           F.cosine_similarity(emb.cpu().double(), want.double(), dim=1).tolist()))
    # softmax sharpness in the oracle: effective number of frames per channel
    h = stages["catconv"]
    a_ = torch.tanh(F.conv1d(h, osd["pooling.linear1.weight"]))
    lg = F.conv1d(a_, osd["pooling.linear2.weight"])
    al = torch.softmax(lg, dim=2)
    print("logits std %.2f max %.2f | effective frames (1/sum alpha^2): median %.1f min %.1f | tanh sat frac %.3f" % (
        lg.std().item(), lg.abs().max().item(), (1 / (al ** 2).sum(2)).median().item(), (1 / (al ** 2).sum(2)).min().item(),
        (a_.abs() > 0.99).float().mean().item()))
