#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` result (rocpd SQLite .db, or *_kernel_stats.csv) as a
small markdown table: per kernel calls / total ms / average us / share, plus grid, LDS and VGPRs."""
import glob
import os
import sqlite3
import sys


def summarise(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
    out = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for name, calls, total, avg, pct in rows:
        out.append("| `%s` | %d | %.3f | %.1f | %.1f |" % (name[:110], calls, total / 1e3, avg, pct))   # top_kernels is in us
    out.append("")
    out.append("| kernel | grid | workgroup | LDS B | VGPR | SGPR | min us | max us |")
    out.append("|---|---|---|---|---|---|---|---|")
    q = ("select name, grid_x, workgroup_x, max(lds_size), max(vgpr_count), max(sgpr_count), min(duration), max(duration) "
         "from kernels group by name, grid_x, workgroup_x order by sum(duration) desc limit 12")
    for name, gx, wx, lds, vg, sg, mn, mx in c.execute(q):
        out.append("| `%s` | %d | %d | %d | %d | %d | %.1f | %.1f |" % (name[:80], gx, wx, lds, vg, sg, mn / 1e3, mx / 1e3))
    # one command runs several workloads through the same instantiation (the default bench.py: cfg3 headline, cfg4 share, parity launches on parts of the
    # batch): the launches of a kernel in clusters of similar duration (a new cluster where the next duration is > 4 % longer), so that a workload's
    # launches can be read off and compared with the average bench.py reports for them
    out.append("")
    out.append("| kernel | launches of similar duration: n x average ms |")
    out.append("|---|---|")
    top = [r[0] for r in rows[:8]]
    for name in top:
        ds = sorted(d for (d,) in c.execute("select duration from kernels where name = ?", (name,)))
        clusters, cur = [], []
        for d in ds:
            if cur and d > 1.04 * cur[-1]:
                clusters.append(cur); cur = []
            cur.append(d)
        if cur:
            clusters.append(cur)
        out.append("| `%s` | %s |" % (name[:80], ", ".join("%d x %.3f" % (len(cl), sum(cl) / len(cl) / 1e6) for cl in clusters if len(cl) >= 2 or len(clusters) <= 4)))
    return "\n".join(out)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        for f in ([p] if p.endswith(".db") else glob.glob(os.path.join(p, "**", "*.db"), recursive=True)):
            print("### %s\n" % os.path.basename(f))
            print(summarise(f))
