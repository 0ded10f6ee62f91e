"""containsAny's early exit, measured: k_sf's time on one document of N bytes whose first KiB matches, against the same document without the
match; next to it the count-mode scan of the same document and (AM_SF_ABLATE=9) the slowest wavefront's own duration, which separates the
launch's fixed cost from the work."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import alfred_margaret_amd as am
from alfred_margaret_amd import synth

def prof(lib, name, fn, reps=5):
    fn()
    am.api.check(lib.am_profile_enable(1)); am.api.check(lib.am_profile_reset())
    for _ in range(reps): fn()
    ms, launches = C.c_double(0), C.c_uint64(0)
    am.api.check(lib.am_profile_read(name, C.byref(ms), C.byref(launches)))
    am.api.check(lib.am_profile_enable(0))
    return ms.value / max(launches.value, 1)

def main():
    needles = synth.needles_for("cfg3_runLower_100k_10GiB")
    a = am.Automaton(needles)
    lib = am.api.libam()
    dev = torch.device("cuda:0")
    needle = needles[0].encode()
    for mib in (1, 16, 64, 256, 1024, 4096):
        n = mib << 20
        text = torch.full((n + 64,), ord("x"), dtype=torch.uint8, device=dev)
        offs = torch.tensor([0, n], dtype=torch.int64, device=dev)
        b = C.c_void_p()
        am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), 1, n, C.byref(b)))
        flags = np.zeros(4, np.uint8)
        counts = np.zeros(1, np.uint64)
        any_fn = lambda: am.api.check(lib.am_contains_any_batch(a.device, am.IGNORE_CASE, b, flags.ctypes.data))
        cnt_fn = lambda: am.api.check(lib.am_count_batch(a.device, am.IGNORE_CASE, b, counts.ctypes.data, None))
        text[100:100 + len(needle)] = torch.tensor(list(needle), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        hit = prof(lib, b"sf", any_fn); f_hit = int(flags[0])
        text[100:100 + len(needle)] = ord("x")
        torch.cuda.synchronize()
        miss = prof(lib, b"sf", any_fn); f_miss = int(flags[0])
        cnt = prof(lib, b"sf", cnt_fn)
        am.api.debug_set("AM_SF_ABLATE", 9)
        cnt_fn()
        out = (C.c_uint64 * 16)()
        lib.am_debug_sf_phase_cycles(out)
        cnt_fn()
        lib.am_debug_sf_phase_cycles(out)
        am.api.debug_set("AM_SF_ABLATE", -1)
        print("%5d MiB: any hit %.4f ms (flag %d)  miss %.4f ms (flag %d)  count %.4f ms;  DBG count: slowest wavefront %.4f ms, %d wavefronts, mean %.4f ms"
              % (mib, hit, f_hit, miss, f_miss, cnt, out[15] * 1e-5, out[4], (out[0] + out[1] + out[2] + out[3]) * 1e-5 / max(out[4], 1)), flush=True)
        lib.am_batch_destroy(b)
main()
