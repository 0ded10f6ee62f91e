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
    return "\n".join(out)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        for f in ([p] if p.endswith(".db") else glob.glob(os.path.join(p, "**", "*.db"), recursive=True)):
            print("### %s\n" % os.path.basename(f))
            print(summarise(f))
