"""Round 6: where the time of the one robustness row below 60 GiB/s goes -- cfg3's needles + the empty needle over cfg3's text (the root's values at almost every
position: k_sf's sparse records + the dense pass) -- per kernel, HIP events inside libam.  usage: empty_needle_times.py [MiB, default 256]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import alfred_margaret_amd as am
from alfred_margaret_amd import synth

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = am.api.libam(); dev = torch.device("cuda:0")
needles = synth.needles_for("cfg3_runLower_100k_10GiB") + [""]
a = am.Automaton(needles)
text, n_bytes = synth.haystacks_device(needles[:-1], True, 0, mib * 1024, dev)
n_hay = mib
offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * (1 << 20)
b = C.c_void_p(); am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(b)))


def run():
    m = C.c_void_p(); am.api.check(lib.am_run_batch(a.device, 1, b, C.byref(m))); n = lib.am_matches_size(m); lib.am_matches_free(m); return n


def count():
    tot = C.c_uint64(0); am.api.check(lib.am_count_batch(a.device, 1, b, None, C.byref(tot))); return tot.value


for name, fn in (("emit", run), ("count", count)):
    fn(); torch.cuda.synchronize()
    am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
    t0 = time.perf_counter()
    for _ in range(3): r = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    am.api.check(lib.am_profile_enable(0))
    parts = []
    for k in (b"sf", b"dense", b"scan", b"permute", b"hidx", b"reduce", b"records_reduce"):
        ms, n = C.c_double(0), C.c_uint64(0); lib.am_profile_read(k, C.byref(ms), C.byref(n))
        if n.value: parts.append("%s %.2f ms x %.1f" % (k.decode(), ms.value / n.value, n.value / 3))
    print("%s: %d, %.1f ms per call = %.1f GiB/s; kernels per call: %s" % (name, r, dt * 1e3, n_bytes / dt / 2**30, ", ".join(parts)), flush=True)
