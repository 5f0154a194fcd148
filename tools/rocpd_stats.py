"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.  Usage: rocpd_stats.py results.db [out.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s "
                  "order by 3 desc" % (name_col, name_col)).fetchall()
total = sum(r[2] for r in rows)
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, c, t, a, mn, mx in rows:
    short = n if len(n) < 110 else n[:107] + "..."
    lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (short, c, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total))
lines.append("| **total** | %d | %.3f | | | | 100 |" % (sum(r[1] for r in rows), total / 1e6))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "a").write(out + "\n")
