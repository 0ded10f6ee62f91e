#!/usr/bin/env python3
"""GPU-box experiment: throughput when haystacks are tiny (many boundaries per 1-KiB chunk)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import alfred_margaret_amd as am
from alfred_margaret_amd import synth
wl = "cfg3_runLower_100k_10GiB"; w = synth.WORKLOADS[wl]
needles = synth.needles_for(wl)
a = am.Automaton(needles)
lib = am.api.libam()
n_cells = 1 << 20   # 1 GiB
text, n_bytes = synth.haystacks_device(needles, True, 0, n_cells, torch.device("cuda:0"))
for hay_bytes in (1 << 20, 4096, 512, 64, 24):
    n_hay = n_bytes // hay_bytes
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device="cuda:0") * hay_bytes
    b = C.c_void_p()
    am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_hay * hay_bytes, C.byref(b)))
    for mode in ("count", "any", "emit"):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            if mode == "count":
                tot = C.c_uint64(0); am.api.check(lib.am_count_batch(a.device, w["case"], b, None, C.byref(tot))); n = tot.value
            elif mode == "any":
                flags = np.zeros(n_hay, np.uint8); am.api.check(lib.am_contains_any_batch(a.device, w["case"], b, flags.ctypes.data)); n = int(flags.sum())
            else:
                m = C.c_void_p(); am.api.check(lib.am_run_batch(a.device, w["case"], b, C.byref(m))); n = lib.am_matches_size(m); lib.am_matches_free(m)
            best = min(best, time.perf_counter() - t0)
        print("haystack %7d B x %9d: %-5s %8.2f ms  %7.1f GiB/s  (result %d)" % (hay_bytes, n_hay, mode, best * 1e3, n_hay * hay_bytes / best / 2**30, n), flush=True)
    lib.am_batch_destroy(b)
