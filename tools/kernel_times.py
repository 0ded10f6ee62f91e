#!/usr/bin/env python3
"""GPU-box experiment: per-kernel times (HIP events inside libam: am_profile_*) of the suffix-filter route on the robustness workloads.
Usage: python tools/kernel_times.py [GiB per case, default 2] [case ...]   cases: cfg3 p0 p8 p64 natural"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import alfred_margaret_amd as am
from alfred_margaret_amd import synth

GIB = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
cases = sys.argv[2:] or ["cfg3", "p64", "natural"]
lib = am.api.libam()
dev = torch.device("cuda:0")
HB = 1 << 20
n_cells = int(GIB * (1 << 20))
cfg3 = synth.needles_for("cfg3_runLower_100k_10GiB")
autos = {}
for case in cases:
    if case == "natural":
        needles = synth.needles_for("natural_100k_10GiB")
        text, n_bytes = synth.haystacks_device(needles, True, 0, n_cells, dev, natural=True)
    else:
        needles = cfg3
        plants = {"cfg3": 1, "p0": 0, "p8": 8, "p64": 64}[case]
        text, n_bytes = synth.haystacks_device(needles, True, 0, n_cells, dev, plants=plants)
    key = id(needles)
    if key not in autos:
        autos[key] = am.Automaton(needles)
    a = autos[key]
    n_hay = n_bytes // HB
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * HB
    batch = C.c_void_p()
    am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(batch)))
    for mode in ("emit", "count"):
        def run():
            if mode == "emit":
                m = C.c_void_p(); am.api.check(lib.am_run_batch(a.device, 1, batch, C.byref(m))); n = int(lib.am_matches_size(m)); lib.am_matches_free(m); return n
            t = C.c_uint64(0); am.api.check(lib.am_count_batch(a.device, 1, batch, None, C.byref(t))); return int(t.value)
        run(); run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): n = run()
        wall = (time.perf_counter() - t0) / 3
        am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
        for _ in range(3): run()
        am.api.check(lib.am_profile_enable(0))
        parts = []
        for k in (b"sf", b"resolve", b"scan", b"permute", b"hidx"):
            ms, cnt = C.c_double(0), C.c_uint64(0)
            am.api.check(lib.am_profile_read(k, C.byref(ms), C.byref(cnt)))
            if cnt.value: parts.append("%s %.3f" % (k.decode(), ms.value / cnt.value))
        print("%-8s %-5s %.2f GiB: wall %.3f ms = %.0f GiB/s; n=%d; kernels (ms/launch): %s" % (case, mode, n_bytes / 2**30, wall * 1e3, n_bytes / 2**30 / wall, n, ", ".join(parts)), flush=True)
    lib.am_batch_destroy(batch)
    del text
