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
text, n_bytes = synth.haystacks_device(needles, w["mixed"], 0, n_hay * cells, torch.device("cuda:0"))
offs = torch.arange(n_hay + 1, dtype=torch.int64, device="cuda:0") * w["hay_bytes"]
lib = am.api.libam()
b = C.c_void_p()
am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(b)))
out = (C.c_uint64 * 11)()
for mode in ("count", "emit"):
    for rep in range(2):
        if mode == "count":
            tot = C.c_uint64(0); am.api.check(lib.am_count_batch(a.device, w["case"], b, None, C.byref(tot)))
        else:
            m = C.c_void_p(); am.api.check(lib.am_run_batch(a.device, w["case"], b, C.byref(m))); lib.am_matches_free(m)
        lib.am_debug_sf_phase_cycles(out)
    waves = max(out[4], 1)
    chunks = n_bytes / 1024 / waves
    tot_c = sum(out[i] for i in range(4))
    nbat = max(out[6], 1)
    print("   resolve batches/wave %.1f, cycles per batch: pre %.0f lookup %.0f preload %.0f walk %.0f" % (out[6] / waves, out[7] / nbat, out[8] / nbat, out[9] / nbat, out[10] / nbat))
    print("%s %s: waves %d, chunks/wave %.0f, cycles/chunk: filter %.0f compact %.0f probe-setup %.0f probe-mem %.0f resolve %.0f  total %.0f" % (
        wl, mode, waves, chunks, out[0] / waves / chunks, out[1] / waves / chunks, out[5] / waves / chunks, out[2] / waves / chunks, out[3] / waves / chunks, (tot_c + out[5]) / waves / chunks))
