#!/usr/bin/env python3
"""GPU-box measurement: Replacer.run on one periodic document (1 MB of "a", needle "aa"): one run of a million overlapping matches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import alfred_margaret_amd as am
r = am.Replacer(0, [("aa", "c")]); r.run("aaaa")
for n in (1 << 20,):
    t0 = time.perf_counter(); out = r.run("a" * n); print(n, "bytes:", round((time.perf_counter() - t0) * 1e3, 2), "ms", len(out), "passes", r.last_stats())
