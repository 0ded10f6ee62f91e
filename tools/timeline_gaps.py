#!/usr/bin/env python3
"""GPU-box aid: from a rocprofv3 --kernel-trace database, how busy each queue was between its first and last kernel, and where the gaps are
(which kernel precedes a gap).  Usage: python tools/timeline_gaps.py <dir with the .db> [min gap us, default 15] [first, last: fractions of the
queue's kernels to look at, default 0.2 1.0]"""
import glob, os, sqlite3, sys
from collections import defaultdict
path = [f for f in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)][0]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
c = sqlite3.connect(path)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")).fetchall()
by_q = defaultdict(list)
for r in rows:
    by_q[r[3] if qcol else 0].append(r)
for q, ks in by_q.items():
    if len(ks) < 50: continue
    # the steady part: drop the first 20 % (set-up, first full scan)
    ks = ks[int(len(ks) * lo):int(len(ks) * hi)]
    span = (ks[-1][2] - ks[0][1]) / 1e3
    busy = sum((k[2] - k[1]) for k in ks) / 1e3
    gaps = defaultdict(lambda: [0, 0.0])
    small = 0.0
    for a, b in zip(ks, ks[1:]):
        g = (b[1] - a[2]) / 1e3
        if g >= min_gap: e = gaps[(a[0][:40], b[0][:40])]; e[0] += 1; e[1] += g
        elif g > 0: small += g
    print("queue %s: %d kernels, span %.1f ms, kernels %.1f ms (%.0f %%), gaps < %.0f us in all %.1f ms" % (q, len(ks), span / 1e3, busy / 1e3, 100 * busy / span, min_gap, small / 1e3))
    durs = defaultdict(lambda: [0, 0.0])
    for k in ks: e = durs[k[0][:50]]; e[0] += 1; e[1] += (k[2] - k[1]) / 1e3
    for name, (n, t) in sorted(durs.items(), key=lambda kv: -kv[1][1])[:24]: print("   kernel %-52s %6d x %6.1f us = %7.1f ms" % (name, n, t / n, t / 1e3))
    for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
        print("   %5d gaps, %.1f ms in all (avg %.0f us): after %s  before %s" % (n, t / 1e3, t / n, a, b))
