"""One rocprofv3 `--kernel-trace --stats` csv -> a markdown table under profiles/.
Usage: python tools/kt_md.py <dir with *_kernel_stats.csv> <profiles/name.md> "<command / workload line>" [passes]"""
import csv
import glob
import os
import re
import sys

src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 60      # 30 preheat + 5 warm-up + 20 timed + 5 stage-bracket passes
hits = glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)
assert len(hits) == 1, hits


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void sugar::", "").replace("sugar::", "")
    return re.sub(r"\(.*$", "", name)[:100]


rows = list(csv.DictReader(open(hits[0])))
lines = ["# rocprofv3 --kernel-trace --stats: %s" % title, "",
         "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
for r in rows:
    lines.append("| `%s` | %s | %.1f | %.1f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                         float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lines += ["", "Kernel time per forward pass: %.4f ms (sum of all kernels / %d passes)." % (tot / 1e6 / passes, passes)]
open(dst, "w").write("\n".join(lines) + "\n")
print(lines[-1])
