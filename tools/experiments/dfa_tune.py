"""Round 6: k_dfa on the natural-text workload under the launch switches of AM_DFA_TUNE (am_dfa.hip dfa_tune): bytes of text a lane asks for at a time,
workgroups per CU, rows kept in LDS.  Per variant: kernel times from HIP events inside libam (am_profile_*), counting and emitting; totals must agree.
usage: dfa_tune.py [GiB] [variant ...]     a variant is name=value (value = the integer AM_DFA_TUNE takes)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import alfred_margaret_amd as am
from alfred_margaret_amd import synth

wl = os.environ.get("AM_TUNE_WORKLOAD", "natural_100k_10GiB")
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
variants = [v.split("=") for v in sys.argv[2:]] or [["base", "0"]]
w = synth.WORKLOADS[wl]
needles = synth.needles_for(wl)
a = am.Automaton(needles); lib = am.api.libam()
dev = torch.device("cuda:0")
n_hay = int(gib * (1 << 30)) // w["hay_bytes"]; cells = w["hay_bytes"] // 1024
text, n_bytes = synth.haystacks_device(needles, w["mixed"], 0, n_hay * cells, dev, natural=bool(w.get("natural")))
offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * w["hay_bytes"]
b = C.c_void_p(); am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(b)))
a.set_kernel(3)


def count():
    tot = C.c_uint64(0); am.api.check(lib.am_count_batch(a.device, w["case"], b, None, C.byref(tot))); return tot.value


def run():
    m = C.c_void_p(); am.api.check(lib.am_run_batch(a.device, w["case"], b, C.byref(m))); n = lib.am_matches_size(m); h = C.c_uint64(0)
    lib.am_matches_free(m); return n


def prof(fn, reps=3):
    fn(); torch.cuda.synchronize()
    am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); am.api.check(lib.am_profile_enable(0))
    out = {}
    for k in (b"dfa", b"dfa_place", b"scan"):
        ms, n = C.c_double(0), C.c_uint64(0); lib.am_profile_read(k, C.byref(ms), C.byref(n)); out[k.decode()] = ms.value / max(1, n.value)
    return r, out


print("workload %s, %.2f GiB, image %d MiB" % (wl, n_bytes / 2**30, len(a.image_bytes(w["case"])) >> 20), flush=True)
ref = None
for name, val in variants:
    am.api.debug_set("AM_DFA_TUNE", int(val, 0))
    c, pc = prof(count)
    r, pr = prof(run)
    if ref is None: ref = (c, r)
    ok = "ok" if (c, r) == ref else "MISMATCH %s vs %s" % ((c, r), ref)
    gi = n_bytes / 2**30
    print("%-22s tune %#8x  count: k_dfa %7.3f ms (%6.1f GiB/s)   emit: k_dfa %7.3f + place %6.3f + scan %5.3f ms (%6.1f GiB/s)   %s" % (
        name, int(val, 0), pc["dfa"], gi / pc["dfa"] * 1e3, pr["dfa"], pr["dfa_place"], pr["scan"], gi / (pr["dfa"] + pr["dfa_place"] + pr["scan"]) * 1e3, ok), flush=True)
am.api.debug_set("AM_DFA_TUNE", -1)
