#!/usr/bin/env python3
"""PCIe-inclusive rate of the one-shot entry points on HOST slices (what a Haskell caller binds): gather into
pinned staging (threads) overlapped with the H2D DMA, scan, results back.  Never the headline `value`."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import alfred_margaret_amd as am
from alfred_margaret_amd import synth

lib = am.api.libam()
needles = synth.needles_for("cfg2_runText_10k_1GiB")
a = am.Automaton(needles)
for n_hay, cells in ((512, 1024), (2048, 1024), (32768, 64)):
    text = synth.haystacks_host(needles, False, 0, n_hay * cells)
    hays = [text[i * cells * 1024:(i + 1) * cells * 1024] for i in range(n_hay)]
    s = am.api._Slices(hays)
    counts = np.zeros(n_hay, np.uint64)
    am.api.check(lib.am_count(a.device, 0, s.arr, s.n, counts.ctypes.data))          # warm-up: staging buffers, image upload
    best_c = best_r = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); am.api.check(lib.am_count(a.device, 0, s.arr, s.n, counts.ctypes.data)); best_c = min(best_c, time.perf_counter() - t0)
        m = C.c_void_p()
        t0 = time.perf_counter(); am.api.check(lib.am_run(a.device, 0, s.arr, s.n, C.byref(m))); p = lib.am_matches_data(m); best_r = min(best_r, time.perf_counter() - t0)
        n_rec = int(lib.am_matches_size(m)); lib.am_matches_free(m)
    print("%6d x %4d KiB = %5d MiB: am_count %.3f s = %.1f GiB/s   am_run (+%d records to host) %.3f s = %.1f GiB/s" %
          (n_hay, cells, text.size >> 20, best_c, text.size / best_c / 2**30, n_rec, best_r, text.size / best_r / 2**30))
