#!/usr/bin/env python3
"""Per-kernel PMC averages from rocprofv3 rocpd .db files (one directory per pass)."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def main(root, kernel_filter="k_sf"):
    acc = defaultdict(lambda: defaultdict(list))
    for db in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
        c = sqlite3.connect(db)
        try:
            cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
            rows = c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall()
        except sqlite3.Error as e:
            print("skip", db, e, file=sys.stderr)
            continue
        per = defaultdict(float)
        for k, cn, v, d in rows:
            per[(k, cn, d)] += v
        for (k, cn, d), v in per.items():
            acc[k][cn].append(v)
    for k in sorted(acc):
        if kernel_filter and kernel_filter not in k:
            continue
        print("### %s" % k[:100])
        for cn in sorted(acc[k]):
            vs = acc[k][cn]
            print("  %-28s n=%d avg=%.6g" % (cn, len(vs), sum(vs) / len(vs)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_sf")
