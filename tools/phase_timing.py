#!/usr/bin/env python3
"""GPU-box experiment: where do k_sf's wavefront cycles go?  (AM_SF_ABLATE=9 enables s_memtime sums.)"""
import ctypes as C, os, sys
os.environ.setdefault("AM_SF_ABLATE", "9")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import alfred_margaret_amd as am
from alfred_margaret_amd import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_runLower_100k_10GiB"
w = synth.WORKLOADS[wl]
needles = synth.needles_for(wl)
a = am.Automaton(needles)
n_hay = 2048 if w["hay_bytes"] >= (1 << 20) else 32768
cells = w["hay_bytes"] // 1024
plants = int(os.environ.get("AM_PLANTS", "1"))
text, n_bytes = synth.haystacks_device(needles, w["mixed"], 0, n_hay * cells, torch.device("cuda:0"), plants=plants, natural=bool(w.get("natural")))
offs = torch.arange(n_hay + 1, dtype=torch.int64, device="cuda:0") * w["hay_bytes"]
lib = am.api.libam()
b = C.c_void_p()
am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(b)))
out = (C.c_uint64 * 16)()
for mode in ("count", "emit"):
    for rep in range(2):
        torch.cuda.synchronize(); import time; t0 = time.perf_counter()
        if mode == "count":
            tot = C.c_uint64(0); am.api.check(lib.am_count_batch(a.device, w["case"], b, None, C.byref(tot)))
        else:
            m = C.c_void_p(); am.api.check(lib.am_run_batch(a.device, w["case"], b, C.byref(m))); lib.am_matches_free(m)
        torch.cuda.synchronize(); call_ms = (time.perf_counter() - t0) * 1e3
        recs = (C.c_uint64 * (2 * 4096))(); lib.am_debug_sf_wave_records(recs, 4096)
        lib.am_debug_sf_phase_cycles(out)
        steps = (C.c_uint64 * 8)(); lib.am_debug_sf_wave_records(steps, 0)
    print("   inside a resolve batch (cycles, no forced waits): until the slot loads are issued %.0f, haystack lookup %.0f, slot line consumed %.0f, walk loop %.0f (%.2f iterations), epilogue (count / emit) %.0f" % (steps[2] / max(out[6], 1), steps[3] / max(out[6], 1), steps[4] / max(out[6], 1), steps[5] / max(out[6], 1), steps[0] / max(out[6], 1), steps[6] / max(out[6], 1)))
    print("   whole call (host clock) %.3f ms" % call_ms)
    import numpy as np
    r = np.frombuffer(recs, dtype=np.uint64).reshape(-1, 2)
    dur = r[:, 0].astype(np.float64); xcc = (r[:, 1] >> np.uint64(32)).astype(np.int64) & 15; hw = (r[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    print("   wave durations: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max()))
    for x in sorted(set(xcc)):
        d = dur[xcc == x]
        print("     xcc %d: waves %d  mean %.0f  min %.0f  max %.0f" % (x, len(d), d.mean(), d.min(), d.max()))
    wg = dur.reshape(-1, 16)
    print("   per workgroup: spread inside a workgroup (max/min) mean %.2f; workgroup means min %.0f max %.0f" % ((wg.max(1) / wg.min(1)).mean(), wg.mean(1).min(), wg.mean(1).max()))
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 7   # gfx9 HW_ID: wave_id 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    for sd in range(4):
        d = dur[simd == sd]
        if len(d): print("     simd %d: waves %d mean %.0f" % (sd, len(d), d.mean()))
    waves = max(out[4], 1)
    chunks = n_bytes / 1024 / waves
    tot_c = sum(out[i] for i in range(4))
    nbat = max(out[6], 1)
    nch = n_bytes / 1024
    print("   slowest wave: %.0f cycles = %.0f per chunk (the averages below are over all waves)" % (out[15], out[15] / (n_bytes / 1024 / max(out[4], 1))))
    print("   per chunk: candidates %.2f, probe batches %.3f, deferred %.2f, resolve batches %.3f, found %.3f" % (out[11] / nch, out[12] / nch, out[13] / nch, out[6] / nch, out[14] / nch))
    print("   resolve batches/wave %.1f, cycles per batch: before the lookup %.0f, lookup + walk %.0f" % (out[6] / waves, out[7] / nbat, (out[8] + out[10]) / nbat))
    print("%s %s: waves %d, chunks/wave %.0f, cycles/chunk: filter %.0f compact %.0f probe-setup %.0f probe-mem %.0f resolve %.0f  total %.0f" % (
        wl, mode, waves, chunks, out[0] / waves / chunks, out[1] / waves / chunks, out[5] / waves / chunks, out[2] / waves / chunks, (out[3] + out[7] + out[8] + out[10]) / waves / chunks, (tot_c + out[5] + out[7] + out[8] + out[10]) / waves / chunks))
