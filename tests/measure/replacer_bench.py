#!/usr/bin/env python3
"""BASELINE config 5 on the GPU box: Replacer.run over a batch of 64-KiB haystacks with 50k (needle, replacement)
pairs, every pass on the device (am_replacer_run_batch).  Prints wall time, passes, bytes scanned and the
per-kernel HIP-event breakdown; checks a sample against the oracle.

  python tests/measure/replacer_bench.py [--n-hay 16384] [--pairs 50000] [--case 0] [--sample 4] [--host-splice-sample 64]
"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import alfred_margaret_amd as am
from alfred_margaret_amd import synth
from oracle import oracle

ap = argparse.ArgumentParser()
ap.add_argument("--n-hay", type=int, default=16384)
ap.add_argument("--hay-kib", type=int, default=64)
ap.add_argument("--pairs", type=int, default=50_000)
ap.add_argument("--case", type=int, default=0)
ap.add_argument("--sample", type=int, default=4)
ap.add_argument("--host-splice-sample", type=int, default=64)
args = ap.parse_args()

dev = torch.device("cuda:0")
lib = am.api.libam()
rng = np.random.default_rng(5)
needles = synth.make_needles(args.pairs, bool(args.case), seed=synth.NEEDLE_SEED + 5)
repls = ["".join(chr(ord("A") + int(x)) for x in rng.integers(0, 26, size=int(rng.integers(0, 17)))) for _ in needles]
pairs = list(zip(needles, repls))       # = synth.replacer_pairs("cfg5_replacer_50k_1GiB") for the default arguments
t0 = time.perf_counter(); r = am.Replacer(args.case, pairs); rdev = C.c_void_p(r.device); build_s = time.perf_counter() - t0

cells = args.hay_kib
n_cells = args.n_hay * cells
text, n_bytes = synth.haystacks_device(needles, bool(args.case), 0, n_cells, dev)
offs = torch.arange(args.n_hay + 1, dtype=torch.int64, device=dev) * (cells * 1024)
batch = C.c_void_p()
am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), args.n_hay, n_bytes, C.byref(batch)))


def run():
    out = C.c_void_p()
    am.api.check(lib.am_replacer_run_batch(rdev, batch, C.c_uint64(2**64 - 1), C.byref(out)))
    return out


res = run(); lib.am_replaced_free(res)          # warm-up: workspaces, pinned staging
am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
torch.cuda.synchronize()
t0 = time.perf_counter(); res = run(); dt = time.perf_counter() - t0
am.api.check(lib.am_profile_enable(0))
passes, scanned = int(lib.am_replaced_passes(res)), int(lib.am_replaced_scanned_bytes(res))
print("Replacer.run on device: %d pairs (build+flatten %.2f s), %d x %d KiB = %.1f MiB, case %d" % (len(pairs), build_s, args.n_hay, cells, n_bytes / 2**20, args.case))
print("  %.3f s -> %.3f GiB/s of input; %d passes, %.2f GiB scanned over all passes (%.1f GiB/s of scanned text)" %
      (dt, n_bytes / dt / 2**30, passes, scanned / 2**30, scanned / dt / 2**30))
tot = 0.0
for k in (b"hidx", b"sf", b"ac", b"scan", b"permute", b"rp_ranges", b"rp_pass", b"rp_scans", b"rp_route", b"rp_splice", b"rp_windows", b"rp_merge"):
    ms, n = C.c_double(0), C.c_uint64(0)
    am.api.check(lib.am_profile_read(k, C.byref(ms), C.byref(n)))
    if n.value:
        print("    %-10s %6d launches %9.2f ms total %8.3f ms avg" % (k.decode(), n.value, ms.value, ms.value / n.value))
        tot += ms.value
print("    kernels total %.1f ms of %.1f ms wall" % (tot, dt * 1e3))


def get(res, i):
    p, n = C.c_void_p(), C.c_size_t(0)
    just = lib.am_replaced_get(res, i, C.byref(p), C.byref(n))
    assert just >= 0
    return C.string_at(p, n.value) if just else None


# parity: oracle on a sample (slow: one CPU scan per pass), scans-on-GPU + host splice on a larger one
host = text[:n_bytes].cpu().numpy()
hays = [bytes(host[i * cells * 1024:(i + 1) * cells * 1024]) for i in range(max(args.sample, args.host_splice_sample))]
t0 = time.perf_counter(); orc = oracle.Replacer(args.case, pairs); exp = [orc.run(h) for h in hays[:args.sample]]; dt_o = time.perf_counter() - t0
assert [get(res, i) for i in range(args.sample)] == exp, "parity failure vs oracle"
print("  parity vs oracle on %d haystacks OK; oracle (1 thread, incl. build) %.2f s for %d KiB -> %.5f GiB/s" %
      (args.sample, dt_o, args.sample * cells, args.sample * cells * 1024 / dt_o / 2**30))
k = args.host_splice_sample
t0 = time.perf_counter(); hs = r.run_batch(hays[:k], host_splice=True); dt_h = time.perf_counter() - t0
assert [get(res, i) for i in range(k)] == hs, "parity failure vs host splice"
print("  parity vs scans-on-GPU + host splice on %d haystacks OK (%.2f s -> %.4f GiB/s)" % (k, dt_h, k * cells * 1024 / dt_h / 2**30))
lib.am_replaced_free(res)
lib.am_batch_destroy(batch)
