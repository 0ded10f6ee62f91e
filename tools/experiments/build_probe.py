"""Round 6: what `build` costs next to `run` (the reference times them together, benchmark/haskell/app/Main.hs:62-64,73): host build (the mirror of Automaton.build),
am_automaton_create (validation = the CaseSensitive flatten; the IgnoreCase flatten on a thread of its own) and the first use (image there? upload), with the flattener's
tasks on and off (AM_FLATTEN_SERIAL).  usage: build_probe.py [workload ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import alfred_margaret_amd as am
from alfred_margaret_amd import synth

lib = am.api.libam()
for wl in sys.argv[1:] or ["cfg3_runLower_100k_10GiB", "natural_100k_10GiB"]:
    if wl.startswith("n="):                       # n=<count>: that many random mixed-case needles, lower-cased, IgnoreCase (the scale check: 1 000 000)
        w = {"case": am.IGNORE_CASE}
        needles = [am.lower_utf8(s).decode("utf-8") for s in synth.make_needles(int(wl[2:]), True)]
    else:
        w = synth.WORKLOADS[wl]
        needles = synth.needles_for(wl)
    for serial in (1, -1, 1, -1):
        am.api.debug_set("AM_FLATTEN_SERIAL", serial)
        t0 = time.perf_counter(); a = am.Automaton(needles); t1 = time.perf_counter()
        n = C.c_size_t(0); am.api.check(lib.am_automaton_image_size(C.c_void_p(a.device), w["case"], C.byref(n))); t2 = time.perf_counter()
        print("%-28s %-8s Automaton() %.3f s + first use of case %d (flatten / wait + upload of %d MiB) %.3f s = build %.3f s" % (
            wl, "serial" if serial == 1 else "tasks", t1 - t0, w["case"], n.value >> 20, t2 - t1, t2 - t0), flush=True)
        del a
am.api.debug_set("AM_FLATTEN_SERIAL", -1)
