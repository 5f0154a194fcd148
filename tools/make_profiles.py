"""Turn one tools/profile_round.sh output directory (gpurun_out/<dir>) into the committed round summaries under profiles/.
Usage: python tools/make_profiles.py gpurun_out/r2p r02"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
pmc = os.path.join(out, tag + "_pmc")
os.makedirs(pmc, exist_ok=True)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void sugar::", "").replace("sugar::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:100]


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern))
    assert len(hits) == 1, (pattern, hits)
    return hits[0]


# ---- kernel trace
rows = list(csv.DictReader(open(one("kt/*/*_kernel_stats.csv"))))
lines = ["# rocprofv3 --kernel-trace --stats, %s" % tag, "",
         "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 "
         "--no-cpu-baseline --no-pipelined` (MI355X, base256 bf16; 30 preheat + 5 warm-up + 20 timed + 5 stage-bracket "
         "single-stream forward passes).", "",
         "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
for r in rows:
    lines.append("| `%s` | %s | %.1f | %.1f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                         float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
passes = 60
lines += ["", "Kernel time per forward pass: %.4f ms (sum of all kernels / %d passes)." % (tot / 1e6 / passes, passes)]
open(os.path.join(out, tag + "_kernel_stats.md"), "w").write("\n".join(lines) + "\n")
shutil.copy(one("kt/*/*_kernel_stats.csv"), os.path.join(out, tag + "_kernel_stats.csv"))

# ---- counter tables
for d, name in (("fetch", "fetch_size"), ("write", "write_size"), ("mfma", "sq_mfma_busy"), ("mfma_wide", "sq_mfma_busy_wide256"),
                ("lds", "sq_wait_lds")):
    f = one(d + "/*/*_counter_collection.csv")
    table = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_table.py"), f]).decode()
    open(os.path.join(pmc, name + "_table.md"), "w").write(table)
    if d in ("fetch", "write"):
        shutil.copy(f, os.path.join(pmc, name + "_counter_collection.csv"))


def means(d):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(one(d + "/*/*_counter_collection.csv"))):
        agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


# ---- HBM-side bytes per launch: FETCH_SIZE (KiB) x 2 on gfx950 for 16-byte-per-lane reads, WRITE_SIZE (KiB) exact
fetch, write = means("fetch"), means("write")
tj = {"_source": "round %s, tools/profile_round.sh (bench.py --steps 3 --warmup 1 --preheat 0, separate FETCH_SIZE / WRITE_SIZE passes)" % tag[1:],
      "_note": "HBM-side bytes per launch = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB x 1024",
      "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    tj["kernels"][k] = {"fetch": int(fetch.get(k, 0) * 1024 * 2), "write": int(write.get(k, 0) * 1024)}
cat = [k for k in tj["kernels"] if k.startswith(("gemm1x1_persist_kernel", "gemm1x1_split_kernel"))]
assert len(cat) == 1
tj["base256_bf16"] = tj["kernels"][cat[0]]["fetch"] + tj["kernels"][cat[0]]["write"]
json.dump(tj, open(os.path.join(out, "traffic_catconv.json"), "w"), indent=1)
print(open(os.path.join(out, tag + "_kernel_stats.md")).read())
print("catconv traffic", tj["base256_bf16"])
