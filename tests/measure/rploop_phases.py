"""Where the cycles of k_rp_loop go on BASELINE config 5 (instrumented instantiation, AM_RP_TRACE=3): per-phase s_memtime sums over all haystacks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["AM_RP_TRACE"] = "3"
import alfred_margaret_amd as am
from alfred_margaret_amd import synth
workload = "cfg5_replacer_50k_1GiB"
w = synth.WORKLOADS[workload]
pairs = synth.replacer_pairs(workload)
n_hay, hb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, w["hay_bytes"]
host = synth.haystacks_host([p[0] for p in pairs], w["mixed"], 0, n_hay * hb // synth.CELL)
hays = [bytes(host[i * hb:(i + 1) * hb]) for i in range(n_hay)]
r = am.Replacer(w["case"], pairs)
r.run_batch(hays[:64])
t0 = time.time()
r.run_batch(hays)
print("%d haystacks, %.1f ms for the call (results copied to the host)" % (n_hay, (time.time() - t0) * 1e3), r.last_stats())
