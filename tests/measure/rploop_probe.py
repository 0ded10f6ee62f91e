"""A few small Replacer calls through the one-kernel loop (csrc/am_rploop.hip) with the host's trace on, under a dead-man timer: the first thing to
run on the GPU after a change to that kernel (a kernel that does not end costs the whole gpurun call)."""
import sys, faulthandler
sys.path.insert(0, ".")
import alfred_margaret_amd as am
from oracle import oracle
am.Automaton(["warm"]).count_matches(0, ["warm up"])
faulthandler.dump_traceback_later(20, exit=True)
am.debug_set("AM_RP_TRACE", 1)
am.debug_set("AM_RP_LOOP", 1)
for case, pairs, hays in ((0, [("aa", "b")], ["aa", "", "aaaa", "xaay"]),
                          (0, [("a", "b"), ("b", "c"), ("c", "dd"), ("dd", "")], ["abcabc" * 50, "", "dddd", "x"]),
                          (1, [("i", "<I>"), ("ß", "ss"), ("k", "K!"), ("å", "")], ["İxİİ", "ẞßẞ", "KkK", "ÅåÅ" * 30, "İẞKÅ" * 100]),
                          (0, [("aa", "b")], ["a" * n for n in (0, 1, 2, 3, 64, 65, 127, 128, 129, 1000, 4097)])):
    got = am.Replacer(case, pairs).run_batch(hays)
    o = oracle.Replacer(case, pairs)
    assert got == [o.run(h) for h in hays], (pairs, got[:3])
    faulthandler.cancel_dump_traceback_later(); faulthandler.dump_traceback_later(20, exit=True)
print("OK", flush=True)
