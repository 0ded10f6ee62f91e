#!/usr/bin/env python3
"""GPU-box check of the two-kernel pipeline (k_filter / k_consume): run with AM_SF_PIPE=1 (forced) and a small slice size so that several
slices, unit chains that cross blocks, far haystacks and the candidate-pool retry all happen on a batch the oracle can check.
Prints "pipeline parity OK" (tests/test_gpu_pipeline.py runs it in a subprocess: the switches are read once per process)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import alfred_margaret_amd as am
from alfred_margaret_amd import synth
from oracle import oracle

assert os.environ.get("AM_SF_PIPE") == "1"
lib = am.api.libam()
dev = torch.device("cuda:0")
needles = synth.needles_for("cfg3_runLower_100k_10GiB")
a = am.Automaton(needles)
table = am.ValuesTable(a)
o = oracle.Machine(needles)
for hay_bytes, n_hay, plants in ((1 << 20, 48, 1), (100 << 10, 300, 8), (3000, 7000, 1)):
    n_cells = hay_bytes * n_hay // 1024
    text, n_bytes = synth.haystacks_device(needles, True, 0, n_cells + 1, dev, plants=plants)
    n_bytes = hay_bytes * n_hay
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * hay_bytes
    batch = C.c_void_p()
    am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(batch)))
    res = {}
    for kernel in (2, 1):
        a.set_kernel(kernel)
        m = C.c_void_p()
        am.api.check(lib.am_run_batch(a.device, 1, batch, C.byref(m)))
        n_rec = int(lib.am_matches_size(m))
        hashes, counts = table.fold_hash(m, n_hay)
        lib.am_matches_free(m)
        per_hay = np.zeros(n_hay, np.uint64); tot = C.c_uint64(0)
        am.api.check(lib.am_count_batch(a.device, 1, batch, per_hay.ctypes.data, C.byref(tot)))
        flags = np.zeros(n_hay, np.uint8)
        am.api.check(lib.am_contains_any_batch(a.device, 1, batch, flags.ctypes.data))
        res[kernel] = (n_rec, hashes, counts, per_hay, int(tot.value), flags)
    sf, ac = res[2], res[1]
    assert sf[0] == ac[0] and np.array_equal(sf[1], ac[1]) and np.array_equal(sf[2], ac[2]), "records differ (pipeline vs general kernel)"
    assert np.array_equal(sf[3], ac[3]) and sf[4] == ac[4] == int(sf[2].sum()), "counts differ"
    assert np.array_equal(sf[5], ac[5]) and np.array_equal(sf[5] != 0, sf[2] != 0), "containsAny differs"
    host = text[:min(n_bytes, 6 * hay_bytes)].cpu().numpy()
    for i in range(min(n_hay, 6)):
        assert o.fold_hash(1, host[i * hay_bytes:(i + 1) * hay_bytes]) == (int(sf[1][i]), int(sf[2][i])), ("oracle", i)
    lib.am_batch_destroy(batch)
    print("shape %d x %d B, %d planted: %d records, %d values" % (n_hay, hay_bytes, plants, sf[0], sf[4]), flush=True)
print("pipeline parity OK")
