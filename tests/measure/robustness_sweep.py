#!/usr/bin/env python3
"""GPU-box experiment: throughput of the match-emitting and the count-only scan away from the one
synthetic point the headline is quoted on.

  density     the cfg3 automaton (100k needles, IgnoreCase) over random text with 0 / 1 / 8 / 64 needles planted per KiB
  natural     Zipf text over a 131k-word vocabulary, needles = 100k vocabulary words (>= 4 letters) and two-word phrases
  a/aa/aaa    needles a, aa, aaa over a...a: a record at EVERY position (16 output bytes per input byte), needles shorter
              than the 4-byte filter window, the record pool's overflow-and-retry path
  empty       the cfg3 needles plus the empty needle: the reference folds the root's values after every successful goto
  concat      haystack = the cfg3 needles written one after the other (a match every ~10 bytes, every filter window hits)

Every case is also a parity check: the fold checksum of the suffix-filter kernel's records equals the general kernel's on
every haystack, and the oracle's on the first haystacks.  Prints a markdown table (profiles/r02_robustness.md).
Usage: python tests/measure/robustness_sweep.py [GiB per case, default 2]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import alfred_margaret_amd as am
am.api.load_check()          # k_ac (set_kernel(1)) is test infrastructure: libam_check.so
from alfred_margaret_amd import synth
from oracle import oracle

GIB = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
lib = am.api.libam()
dev = torch.device("cuda:0")
HB = 1 << 20


def measure(name, needles, case, text, n_bytes, hay_bytes=HB, steps=4, oracle_hays=8, note=""):
    n_hay = n_bytes // hay_bytes
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * hay_bytes
    a = am.Automaton(needles)
    table = am.ValuesTable(a)
    batch = C.c_void_p()
    am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(batch)))
    res = {}
    try:
        def run(kernel, keep=False):
            a.set_kernel(kernel)
            m = C.c_void_p()
            am.api.check(lib.am_run_batch(a.device, case, batch, C.byref(m)))
            n = int(lib.am_matches_size(m))
            out = table.fold_hash(m, n_hay) if keep else None
            lib.am_matches_free(m)
            return n, out

        def count(kernel):
            a.set_kernel(kernel)
            t = C.c_uint64(0)
            am.api.check(lib.am_count_batch(a.device, case, batch, None, C.byref(t)))
            return int(t.value)

        for kernel, tag in ((0, "auto"), (2, "sf"), (1, "ac")):
            try:
                n_rec, (hashes, counts) = run(kernel, keep=True)      # also the warm-up (pool sizing, image upload)
            except am.AmError as e:
                if e.code != am.AM_ERR_UNSUPPORTED:
                    raise
                res[tag] = None
                continue
            reps = steps if tag != "ac" else 1
            am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                run(kernel)
            emit_s = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                total = count(kernel)
            count_s = (time.perf_counter() - t0) / reps
            am.api.check(lib.am_profile_enable(0))
            ms, nl = C.c_double(0), C.c_uint64(0)
            am.api.check(lib.am_profile_read(b"dfa", C.byref(ms), C.byref(nl)))
            res[tag] = dict(emit=n_bytes / 2**30 / emit_s, count=n_bytes / 2**30 / count_s, records=n_rec, matches=total, hashes=hashes, counts=counts, route="k_dfa" if nl.value else "k_sf")
        if res.get("sf") and res.get("ac"):
            assert np.array_equal(res["sf"]["hashes"], res["ac"]["hashes"]) and np.array_equal(res["sf"]["counts"], res["ac"]["counts"]), name
        if res.get("auto") and res.get("ac"):
            assert np.array_equal(res["auto"]["hashes"], res["ac"]["hashes"]) and np.array_equal(res["auto"]["counts"], res["ac"]["counts"]), name
        ref = res.get("sf") or res["ac"]
        o = oracle.Machine(needles)
        k = min(oracle_hays, n_hay)
        host = text[:k * hay_bytes].cpu().numpy()
        for i in range(k):
            assert o.fold_hash(case, host[i * hay_bytes:(i + 1) * hay_bytes]) == (int(ref["hashes"][i]), int(ref["counts"][i])), (name, i)
    finally:
        lib.am_batch_destroy(batch)
    kib = n_bytes / 1024.0
    sf, ac, au = res.get("sf"), res.get("ac"), res.get("auto")
    fmt = lambda r, key: "%.0f" % r[key] if r else "n/a"
    print("| %s | %.1f | %.2f | %s | %s | %s | %s | %s | %s | %s | %s |" % (name, (ref["matches"] / kib), ref["records"] / kib, au["route"] if au else "n/a", fmt(au, "emit"), fmt(au, "count"),
                                                                  fmt(sf, "emit"), fmt(sf, "count"), fmt(ac, "emit"), fmt(ac, "count"), note), flush=True)


def main():
    print("| case | matches / KiB | records / KiB | the library's route | emit GiB/s | count GiB/s | k_sf emit | k_sf count | k_ac emit | k_ac count | note |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    n_cells = int(GIB * (1 << 20))
    cfg3 = synth.needles_for("cfg3_runLower_100k_10GiB")
    for plants in (0, 1, 8, 64):
        text, n_bytes = synth.haystacks_device(cfg3, True, 0, n_cells, dev, plants=plants)
        measure("random text, %d needles planted per KiB" % plants, cfg3, 1, text, n_bytes, note="BASELINE workload" if plants == 1 else "")
        del text
    nat = synth.needles_for("natural_100k_10GiB")
    text, n_bytes = synth.haystacks_device(nat, True, 0, n_cells, dev, natural=True)
    measure("natural text (Zipf vocabulary), 100k vocabulary needles", nat, 1, text, n_bytes, oracle_hays=4)
    del text
    # needles from the natural vocabulary, text that never contains them (the same filter, no matches)
    text, n_bytes = synth.haystacks_device(cfg3, True, 0, n_cells, dev, plants=0)
    measure("natural needles over random text", nat, 1, text, n_bytes, oracle_hays=4)
    del text
    small = max(64, n_cells // 8) * 1024                          # a record per position: 16 B out per byte in
    text = torch.full((small + 64,), ord("a"), dtype=torch.uint8, device=dev)
    text[small:] = 0
    measure("a, aa, aaa over a...a", ["a", "aa", "aaa"], 0, text, small, oracle_hays=2, note="a record at every position; %d MiB" % (small >> 20))
    del text
    text, n_bytes = synth.haystacks_device(cfg3, True, 0, small // 1024, dev)
    measure("cfg3 needles + the empty needle, random text", cfg3 + [""], 1, text, n_bytes, oracle_hays=2,
            note="the root's values at (almost) every position: k_sf + dense pass; %d MiB" % (small >> 20))
    del text
    blob = ("".join(cfg3)).encode("utf-8")                       # ~1 MB: one haystack = its first MiB (cut at a code-point boundary, padded), repeated
    one = np.frombuffer((blob * (HB // len(blob) + 1))[:HB], dtype=np.uint8).copy()
    cut = HB
    while cut > 0 and (one[cut - 1] & 0xC0) == 0x80:
        cut -= 1
    if cut > 0 and one[cut - 1] >= 0xC0:
        cut -= 1
    one[cut:] = ord(" ")
    text = torch.from_numpy(np.concatenate([np.tile(one, small // HB), np.zeros(64, np.uint8)])).to(dev)
    measure("haystack = the 100k needles concatenated", cfg3, 1, text, small, oracle_hays=2, note="%d MiB" % (small >> 20))


if __name__ == "__main__":
    main()
