#!/usr/bin/env python3
"""GPU-box A/B: k_sf vs the role-specialised k_sfx on one device-resident batch (same process, same batch, alternating), kernel time from
libam's HIP-event profile, plus k_sfx's per-role cycle sums (AM_SF_ABLATE=9 instantiation).
usage: sfx_ab.py [workload] [GiB] [reps]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import alfred_margaret_amd as am
from alfred_margaret_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_runLower_100k_10GiB"
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = synth.WORKLOADS[wl]
needles = synth.needles_for(wl)
a = am.Automaton(needles)
n_hay = int(gib * (1 << 30)) // w["hay_bytes"]
cells = w["hay_bytes"] // 1024
plants = int(os.environ.get("AM_PLANTS", "1"))
text, n_bytes = synth.haystacks_device(needles, w["mixed"], 0, n_hay * cells, torch.device("cuda:0"), plants=plants, natural=bool(w.get("natural")))
offs = torch.arange(n_hay + 1, dtype=torch.int64, device="cuda:0") * w["hay_bytes"]
lib = am.api.libam()
b = C.c_void_p()
am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(b)))


def run(mode):
    if mode == "count":
        tot = C.c_uint64(0); am.api.check(lib.am_count_batch(a.device, w["case"], b, None, C.byref(tot))); return int(tot.value)
    m = C.c_void_p(); am.api.check(lib.am_run_batch(a.device, w["case"], b, C.byref(m)))
    n = int(lib.am_matches_size(m)); lib.am_matches_free(m); return n


def kernel_ms(sfx, mode, n):
    am.debug_set("AM_SFX", sfx)
    run(mode)
    am.api.check(lib.am_profile_enable(1)); am.api.check(lib.am_profile_reset())
    t0 = time.perf_counter()
    for _ in range(n):
        res = run(mode)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3 / n
    ms, launches = C.c_double(0), C.c_uint64(0)
    am.api.check(lib.am_profile_read(b"sf", C.byref(ms), C.byref(launches)))
    am.api.check(lib.am_profile_enable(0))
    return ms.value / max(launches.value, 1), wall, res


print("%s: %d haystacks, %.2f GiB, %d needles, plants %d" % (wl, n_hay, n_bytes / 2**30, len(needles), plants))
for mode in ("emit", "count"):
    rows = []
    for rnd in range(2):
        for sfx in (0, 1):
            k, wall, res = kernel_ms(sfx, mode, reps)
            rows.append((sfx, k, wall, res))
    for sfx in (0, 1):
        ks = [r[1] for r in rows if r[0] == sfx]; ws = [r[2] for r in rows if r[0] == sfx]; rs = {r[3] for r in rows if r[0] == sfx}
        print("  %-5s %-5s kernel %.3f / %.3f ms (%.0f GiB/s), call %.3f ms, result %s" % (mode, "k_sfx" if sfx else "k_sf", ks[0], ks[1], n_bytes / 2**30 / (min(ks) * 1e-3), min(ws), sorted(rs)))
    assert len({r[3] for r in rows}) == 1, rows

# per-role cycle sums of k_sfx (instrumented instantiation)
am.debug_set("AM_SFX", 1); am.debug_set("AM_SF_ABLATE", 9)
fn = lib.am_debug_sfx_roles; fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
out = (C.c_uint64 * 24)()
for mode in ("emit",):
    run(mode); am.api.check(fn(out))
    t0 = time.perf_counter(); run(mode); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    am.api.check(fn(out))
    r = [int(x) for x in out]
    info = am.device_info(); wgs = info["n_cu"]
    chunks = max(r[3], 1)
    nF, nP = 12 * wgs, 4 * wgs
    print("  k_sfx roles (%s, instrumented, call %.2f ms): per F wavefront %.0f cycles = %.0f per chunk; waiting for ring room %.1f %%, for a unit slot %.1f %%; candidates per chunk %.1f" % (
        mode, wall, r[0] / nF, r[0] / chunks, 100.0 * r[1] / max(r[0], 1), 100.0 * r[2] / max(r[0], 1), r[4] / chunks))
    print("    F resolve: batches %d with %.1f positions each, %.0f cycles per batch = %.1f %% of the F time; found %d" % (
        r[18], r[19] / max(r[18], 1), r[21] / max(r[18], 1), 100.0 * r[21] / max(r[0], 1), r[20]))
    print("    per P wavefront %.0f cycles; passes %d (%.0f cycles each), idle %.1f %%, rounds %d with %.1f entries each; waiting for q2 room %.1f %%; deferred per chunk %.2f" % (
        r[8] / nP, r[10], r[8] / max(r[10], 1), 100.0 * r[11] / max(r[10], 1), r[12], r[13] / max(r[12], 1), 100.0 * r[9] / max(r[8], 1), r[14] / chunks))
am.debug_set("AM_SF_ABLATE", -1)
lib.am_batch_destroy(b)
