"""Round 6: am_run on HOST slices of the natural-text workload (a result 2.5 x the size of its text): where the time of a call goes -- am_run (gather + upload + scan) and
am_matches_data (the records' way back) timed apart, three calls.  usage: run_host_natural.py [GiB]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import alfred_margaret_amd as am
from alfred_margaret_amd import synth

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
wl = "natural_100k_10GiB"; w = synth.WORKLOADS[wl]
needles = synth.needles_for(wl)
a = am.Automaton(needles); lib = am.api.libam()
cells = w["hay_bytes"] // 1024; n_hay = int(gib * (1 << 30)) // w["hay_bytes"]
text = synth.haystacks_host(needles, w["mixed"], 0, n_hay * cells, natural=True)
hays = [text[i * cells * 1024:(i + 1) * cells * 1024] for i in range(n_hay)]
s = am.api._Slices(hays)
lib.am_matches_data.restype = C.c_void_p
for rep in range(4):
    m = C.c_void_p()
    t0 = time.perf_counter(); am.api.check(lib.am_run(a.device, w["case"], s.arr, s.n, C.byref(m))); t1 = time.perf_counter()
    p = lib.am_matches_data(m); t2 = time.perf_counter()
    n = int(lib.am_matches_size(m))
    rec = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n * 2,))
    chk = int(rec[0::2][:: max(1, n // 1000)].sum())          # touch the result
    lib.am_matches_free(m); t3 = time.perf_counter()
    print("call %d: am_run %.1f ms, am_matches_data %.1f ms (%d records = %.2f GB), free %.1f ms: %.1f GiB/s of text end to end   [%x]" % (
        rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n, n * 16 / 1e9, (t3 - t2) * 1e3, gib / (t2 - t0), chk & 0xFFFF), flush=True)
