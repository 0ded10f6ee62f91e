#!/usr/bin/env python3
"""Latency of one-shot calls on small inputs (BASELINE configs[0]: 3 needles, one 1-MB ASCII haystack; and a 100-byte
haystack): what a caller that scans one document per call pays.  Compared with the CPU oracle on the same input."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import alfred_margaret_amd as am
from oracle import oracle

lib = am.api.libam()
needles = ["tshirt", "shirts", "shorts"]
a = am.Automaton(needles)
o = oracle.Machine(needles)
rng = np.random.default_rng(1)
words = ["short", "tshirts", "sweatshirts", "and", "shirtshirts", "the", "quick", "brown", "fox", "shorts"]
big = (" ".join(words[int(i)] for i in rng.integers(0, len(words), size=200000)))[:1_000_000].encode()
for name, hay in (("1 MB", big), ("100 B", big[:100]), ("10 KB", big[:10_000])):
    s = am.api._Slices([hay])
    counts = np.zeros(1, np.uint64)
    flags = np.zeros(1, np.uint8)
    for _ in range(3):
        am.api.check(lib.am_count(a.device, 0, s.arr, 1, counts.ctypes.data))
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        am.api.check(lib.am_count(a.device, 0, s.arr, 1, counts.ctypes.data))
    t_count = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        am.api.check(lib.am_contains_any(a.device, 0, s.arr, 1, flags.ctypes.data))
    t_any = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        m = C.c_void_p(); am.api.check(lib.am_run(a.device, 0, s.arr, 1, C.byref(m))); lib.am_matches_data(m); lib.am_matches_free(m)
    t_run = (time.perf_counter() - t0) / n
    t0 = time.perf_counter(); c = o.count_matches(0, hay); t_cpu = time.perf_counter() - t0
    assert int(counts[0]) == c
    print("%6s haystack: am_count %7.1f us  am_contains_any %7.1f us  am_run %7.1f us   CPU oracle count %9.1f us (%d matches)" %
          (name, t_count * 1e6, t_any * 1e6, t_run * 1e6, t_cpu * 1e6, c))

# Replacer.run on one small document per call
pairs = [("tshirt", "T-SHIRT"), ("shorts", "pants"), ("quick", "slow"), ("fox", "dog")]
r = am.Replacer(0, pairs)
orc = oracle.Replacer(0, pairs)
for name, hay in (("100 B", big[:100]), ("10 KB", big[:10_000]), ("1 MB", big)):
    for _ in range(3):
        out = r.run(hay)
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        out = r.run(hay)
    t_gpu = (time.perf_counter() - t0) / n
    t0 = time.perf_counter(); exp = orc.run(hay); t_cpu = time.perf_counter() - t0
    assert out == exp
    print("%6s document: Replacer.run %8.1f us (%d passes)   CPU oracle %9.1f us" % (name, t_gpu * 1e6, r.last_stats()[0], t_cpu * 1e6))
