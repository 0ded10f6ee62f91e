"""Round 6 (CPU only): a model of what one XCD's L2 sees of k_dfa on the natural-text workload -- requests and misses by what was asked for (text, chain records, hot
table, cold columns) -- to see which table a layout change should shrink.  usage: dfa_l2sim.py [lanes] [unit] [l2 MiB]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import alfred_margaret_amd as am
from alfred_margaret_amd import synth
from tests.helpers import ImgCheck

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
unit = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
l2 = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
wl = "natural_100k_10GiB"; w = synth.WORKLOADS[wl]
needles = synth.needles_for(wl)
chk = ImgCheck()
for k, v in (("AM_DFA_HOT_LOG2", os.environ.get("AM_DFA_HOT_LOG2")),):
    if v: chk.set(k, int(v))
img = chk.flatten(am.Automaton(needles), w["case"])
text = np.frombuffer(synth.haystacks_host(needles, w["mixed"], 0, lanes * unit // 1024, natural=True), dtype=np.uint8)
out = np.zeros(12, np.uint64)
chk.lib.amchk_dfa_l2sim.restype = C.c_longlong
rc = chk.lib.amchk_dfa_l2sim(img.ctypes.data_as(C.c_void_p), text.ctypes.data_as(C.c_void_p), C.c_uint64(lanes), C.c_uint32(unit), C.c_uint32(512), C.c_uint64(int(l2 * (1 << 20))), C.c_uint32(16), C.c_uint32(int(os.environ.get("SIM_HOT16", "0"))),
                             out.ctypes.data_as(C.c_void_p))
assert rc == 0
steps = lanes * unit
names = ("text (64-byte pieces)", "chain records", "hot table", "cold columns", "rare-byte walk", "LDS / no load")
print("%d lanes x %d bytes, L2 %.1f MiB: per step" % (lanes, unit, l2))
for c, nm in enumerate(names):
    print("  %-22s requests %.4f  misses %.4f" % (nm, out[2 * c] / steps, out[2 * c + 1] / steps))
print("  %-22s requests %.4f  misses %.4f" % ("all", out[0:10:2].sum() / steps, out[1:10:2].sum() / steps))
