"""Round 5: first timing of k_dfa against k_sf on the natural-text workload (and on cfg3 with AM_DFA=1)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import alfred_margaret_amd as am
from alfred_margaret_amd import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "natural_100k_10GiB"
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
w = synth.WORKLOADS[wl]
needles = synth.needles_for(wl)
all_needles = needles
if os.environ.get("AM_PROBE_NEEDLES"): needles = needles[:int(os.environ["AM_PROBE_NEEDLES"])]
t0 = time.time(); a = am.Automaton(needles); lib = am.api.libam()
dev = torch.device("cuda:0")
n_hay = int(gib * (1 << 30)) // w["hay_bytes"]; cells = w["hay_bytes"] // 1024
text, n_bytes = synth.haystacks_device(all_needles, w["mixed"], 0, n_hay * cells, dev, natural=bool(w.get("natural")))
offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * w["hay_bytes"]
b = C.c_void_p(); am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(b)))
def timed(label, k, fn, reps=3):
    a.set_kernel(k); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("%-28s %8.2f ms  %7.1f GiB/s   result %s" % (label, min(ts) * 1e3, n_bytes / float(1 << 30) / min(ts), r), flush=True)
def count():
    tot = C.c_uint64(0); am.api.check(lib.am_count_batch(a.device, w["case"], b, None, C.byref(tot))); return tot.value
def run():
    m = C.c_void_p(); am.api.check(lib.am_run_batch(a.device, w["case"], b, C.byref(m))); n = lib.am_matches_size(m); lib.am_matches_free(m); return n
def anyf():
    import numpy as np
    f = np.zeros(n_hay, np.uint8); am.api.check(lib.am_contains_any_batch(a.device, w["case"], b, f.ctypes.data)); return int(f.sum())
timed("build+first", 2, count, 1); print("build %.2f s, image %d MiB" % (time.time() - t0, len(a.image_bytes(w["case"])) >> 20))
for name, k in (("sf", 2), ("dfa", 3)):
    timed(name + " count", k, count); timed(name + " emit", k, run); timed(name + " any", k, anyf)
am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1)); a.set_kernel(3); run(); torch.cuda.synchronize(); am.api.check(lib.am_profile_enable(0))
for k in (b"dfa", b"dfa_place", b"scan", b"hidx"):
    ms, n = C.c_double(0), C.c_uint64(0); lib.am_profile_read(k, C.byref(ms), C.byref(n)); print(k.decode(), round(ms.value, 3), "ms in", n.value, "launches")
