"""
ORACLE -- TEST INFRASTRUCTURE / ANALYSIS ONLY (never imported by the product path).

Which tensor's bf16 rounding moves the bf16 path's error on the ill-conditioned weight fill?  

The f32 oracle forward (oracle/ref_cpu.py::_subnet_forward, restating dynamictdnn.py:645-666) is re-run with a bf16
round-trip injected at the places where the GPU's bf16 path stores a bf16 tensor: the features,
every conv weight (by layer group: w:layer1 ... w:pool2), layer1's output / the running sum S, first_conv's output, the Res2 step inputs u = x + sp and outputs
sp, last_conv's output y, the block output, the catconv output h and the tanh output a.  Everything else (accumulation,
folded BN, SE, softmax and pooled statistics, the tail) stays f32, as on the GPU.  For each class the table gives the
error (max over utterances of the relative L2 distance to the f32 embedding, and the min cosine) with ONLY that class
rounded and with ALL BUT that class rounded.

    python3 -W ignore oracle/bf16_ablation.py [t ...]      # t: position on synth.conditioning's path, default 0 1

Writes profiles/r03_bf16_ablation.md.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import ref_cpu  # noqa: E402
from oracle.ref_cpu import SCALE, _bn, _conv  # noqa: E402
from sugar_amd import layout, synth  # noqa: E402

W_CLASSES = ["w:layer1", "w:first", "w:res2", "w:last", "w:catconv", "w:pool1", "w:pool2"]
CLASSES = ["feats"] + W_CLASSES + ["S", "first", "res2_u", "res2_sp", "y", "out", "h", "a"]


def _wclass(key):
    for tag, name in (("layer1", "w:layer1"), ("first_conv", "w:first"), ("resconv", "w:res2"), ("last_conv", "w:last"),
                      ("catconv", "w:catconv"), ("linear1", "w:pool1"), ("linear2", "w:pool2")):
        if tag in key:
            return name
    raise KeyError(key)


def forward(sd, depth, feats, on):
    """_subnet_forward with bf16 round trips at the tensor classes in `on`."""
    def r(name, t):
        return t.to(torch.bfloat16).to(torch.float32) if name in on else t

    def w(key):
        return r(_wclass(key), sd[key])

    def crb(x, prefix, dil=1):
        return _bn(F.relu(_conv(x, w(prefix + ".conv.weight"), dil)), sd, prefix + ".bn")

    with torch.no_grad():
        out = r("S", crb(r("feats", feats), "layer1"))
        cat_out, sum_out = [], out
        for i in range(depth):
            p = "blocks.%d." % i
            h = r("first", crb(out, p + "first_conv"))
            spx = torch.split(h, h.shape[1] // SCALE, 1)
            outs, sp = [], None
            for j in range(SCALE - 1):
                q = p + "resconv.convs.%d" % j
                inp = spx[j] if j == 0 else r("res2_u", spx[j] + sp)
                sp = r("res2_sp", _bn(F.relu(_conv(inp, w(q + ".conv.weight"), i + 2)), sd, q + ".bn"))
                outs.append(sp)
            outs.append(spx[SCALE - 1])
            h = r("y", crb(torch.cat(outs, dim=1), p + "last_conv"))
            s = h.mean(dim=2)
            s = F.relu(F.linear(s, sd[p + "se.reduce.weight"], sd[p + "se.reduce.bias"]))
            s = torch.sigmoid(F.linear(s, sd[p + "se.expand.weight"], sd[p + "se.expand.bias"]))
            out = r("out", h * s.unsqueeze(2) + sum_out)
            cat_out.append(out)
            sum_out = r("S", sum_out + out)
        h = r("h", F.relu(_conv(torch.cat(cat_out, dim=1), w("catconv.weight"))))
        a = r("a", torch.tanh(_conv(h, w("pooling.linear1.weight"))))
        a = torch.softmax(_conv(a, w("pooling.linear2.weight")), dim=2)
        mean = torch.sum(a * h, dim=2)
        std = torch.sqrt((torch.sum(a * h ** 2, dim=2) - mean ** 2).clamp(min=1e-9))
        z = _bn(torch.cat([mean, std], dim=1), sd, "bnpool")
        z = F.linear(z, sd["linear.weight"], sd["linear.bias"])
        return _bn(z, sd, "bnembed")


def err(e, want):
    e, want = e.double(), want.double()
    return ((e - want).norm(dim=1) / want.norm(dim=1)).max().item(), F.cosine_similarity(e, want, dim=1).min().item()


def main():
    ts = [float(x) for x in sys.argv[1:]] or [0.0, 1.0]
    torch.set_num_threads(8)
    wav = synth.waveforms(4, 48000, seed=0)
    feats = ref_cpu.logmel_fbank(wav).squeeze(1)
    lines = ["# bf16 error budget of the embedding path (oracle/bf16_ablation.py): CPU emulation, 4 x 3 s utterances.",
             "# rel = max over utterances of |e - e_f32| / |e_f32|, cos = min cosine; 'only' = this tensor class alone is",
             "# rounded to bf16, 'all but' = every class except this one.", ""]
    for name, cfg in (("base", layout.BASE), ("largest", layout.LARGEST)):
        for t in ts:
            sd = ref_cpu.clone_state(synth.fill_state_dict(layout.supernet_shapes(), seed=0, **synth.conditioning(t)), cfg)
            want = forward(sd, cfg[0], feats, set())
            assert torch.equal(want, ref_cpu.subnet_forward(sd, cfg[0], feats))       # the restatement IS the oracle
            rel_all, cos_all = err(forward(sd, cfg[0], feats, set(CLASSES)), want)
            lines += ["## %s, t = %.3f (t = 0 shipped fill, t = 1 ill-conditioned): all classes rounded: rel %.2e, cos %.6f"
                      % (name, t, rel_all, cos_all), "", "| class | only: rel | only: cos | all but: rel | all but: cos |",
                      "|---|---|---|---|---|"]
            for c in CLASSES:
                r1, c1 = err(forward(sd, cfg[0], feats, {c}), want)
                r2, c2 = err(forward(sd, cfg[0], feats, set(CLASSES) - {c}), want)
                lines.append("| %s | %.2e | %.6f | %.2e | %.6f |" % (c, r1, c1, r2, c2))
            lines.append("")
            print("\n".join(lines[-(len(CLASSES) + 5):]), flush=True)
    with open(os.path.join(ROOT, "profiles", "r03_bf16_ablation.md"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
