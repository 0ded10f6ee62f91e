"""k_sfx (csrc/am_sfx.hip): the suffix-filter scan with role-specialised wavefronts (12 filter, 3 probe, 1 resolve per workgroup) must be
bit-identical to k_sf on every batch shape -- same records in the same order (Automaton.hs:442-534: position order per haystack), same
counts -- and both to the oracle.  AM_SFX=1 forces it on batches far smaller than the ones it is chosen for by default, so that ragged
shapes, every unit size (1 .. 64 chunks), empty haystacks, the pool-overflow retry and the CaseSensitive image are all exercised."""
import ctypes as C
import functools

import numpy as np
import pytest

import alfred_margaret_amd as am
from alfred_margaret_amd import synth
from oracle import oracle
from tests.helpers import expand_records

pytestmark = pytest.mark.gpu


@functools.lru_cache(maxsize=1)
def _cfg3():
    needles = synth.needles_for("cfg3_runLower_100k_10GiB")
    return needles, am.Automaton(needles), oracle.Machine(needles)


def _batch_from_device(text, offs, n_hay, n_bytes):
    b = C.c_void_p()
    am.api.check(am.libam().am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(b)))
    return b


def _run(a, case, b):
    m = C.c_void_p()
    am.api.check(am.libam().am_run_batch(a.device, case, b, C.byref(m)))
    try:
        return am.api.matches_to_numpy(m)
    finally:
        am.libam().am_matches_free(m)


def _count(a, case, b, n_hay):
    counts = np.zeros(n_hay, np.uint64)
    total = C.c_uint64(0)
    am.api.check(am.libam().am_count_batch(a.device, case, b, counts.ctypes.data, C.byref(total)))
    return counts, int(total.value)


def _ragged_offsets(rng, n_bytes, n_hay):
    cuts = np.sort(rng.integers(0, n_bytes + 1, size=n_hay - 1))
    offs = np.concatenate([[0], cuts, [n_bytes]]).astype(np.int64)
    # a few empty haystacks and one that ends a few bytes after a 1-KiB boundary
    offs[5] = offs[4]
    if n_hay > 9:
        offs[9] = offs[8]
    return np.maximum.accumulate(offs)


@pytest.mark.parametrize("mib,n_hay", [(2, 7), (24, 96), (160, 300)])
def test_sfx_equals_sf_equals_oracle(mib, n_hay):
    import torch
    needles, a, o = _cfg3()
    dev = torch.device("cuda:0")
    text, n_bytes = synth.haystacks_device(needles, True, 17 * mib, mib * (1 << 20) // synth.CELL, dev)
    n_bytes -= 37                                            # the batch does not end on a chunk boundary
    rng = np.random.default_rng(mib)
    offs_h = _ragged_offsets(rng, n_bytes, n_hay)
    offs = torch.from_numpy(offs_h).to(dev)
    b = _batch_from_device(text, offs, n_hay, n_bytes)
    try:
        host = text[:n_bytes].cpu().numpy()
        for case in (am.IGNORE_CASE, am.CASE_SENSITIVE):
            am.debug_set("AM_SFX", 0)
            ref = _run(a, case, b)
            ref_counts, ref_total = _count(a, case, b, n_hay)
            am.debug_set("AM_SFX", 1)
            got = _run(a, case, b)
            got_counts, got_total = _count(a, case, b, n_hay)
            assert len(got) == len(ref) and got.tobytes() == ref.tobytes(), (case, len(got), len(ref))
            assert got_total == ref_total and np.array_equal(got_counts, ref_counts)
            assert len(got) > (mib * 1024 if case else mib * 20)
            # the oracle on a spread sample of haystacks: full (matchPos, value) fold sequences
            first = np.searchsorted(got["haystack"], np.arange(n_hay + 1))
            for h in list(range(0, n_hay, max(1, n_hay // 12)))[:12]:
                seg = host[offs_h[h]:offs_h[h + 1]]
                if len(seg) > (4 << 20):
                    continue
                pos, val = o.run_list(case, seg)
                rs = got[first[h]:first[h + 1]]
                assert expand_records(o.values_off(), o.values(), rs["haystack"], rs["state"], rs["end_pos"]) == [(h, int(p), int(v)) for p, v in zip(pos, val)], (case, h)
                assert int(got_counts[h]) == len(pos)
    finally:
        am.libam().am_batch_destroy(b)


def test_sfx_pool_overflow_retry_and_repeatability():
    import torch
    needles, a, _ = _cfg3()
    dev = torch.device("cuda:0")
    mib, n_hay = 48, 48
    text, n_bytes = synth.haystacks_device(needles, True, 3, mib * (1 << 20) // synth.CELL, dev)
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * (n_bytes // n_hay)
    n_bytes = int(offs[-1].item())
    b = _batch_from_device(text, offs, n_hay, n_bytes)
    try:
        am.debug_set("AM_SFX", 0)
        ref = _run(a, am.IGNORE_CASE, b)
        am.debug_set("AM_SFX", 1)
        am.debug_set("AM_SF_POOL_BLOCKS", 40)               # far too small: the kernel keeps counting, the host retries with the exact size
        got = _run(a, am.IGNORE_CASE, b)
        am.debug_set("AM_SF_POOL_BLOCKS", -1)
        assert got.tobytes() == ref.tobytes()
        for _ in range(5):                                   # the hand-over between the wavefronts is timing dependent; the result is not
            assert _run(a, am.IGNORE_CASE, b).tobytes() == ref.tobytes()
    finally:
        am.libam().am_batch_destroy(b)


def test_sfx_role_counters_debug_build():
    """AM_SF_ABLATE=9 runs the instrumented instantiation; its per-role counters must add up (chunks filtered = the batch, candidates
    popped = candidates pushed, positions resolved = positions deferred) and the records must not change."""
    import torch
    needles, a, _ = _cfg3()
    dev = torch.device("cuda:0")
    mib, n_hay = 64, 64
    text, n_bytes = synth.haystacks_device(needles, True, 11, mib * (1 << 20) // synth.CELL, dev)
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * (n_bytes // n_hay)
    b = _batch_from_device(text, offs, n_hay, n_bytes)
    try:
        am.debug_set("AM_SFX", 1)
        ref = _run(a, am.IGNORE_CASE, b)
        am.debug_set("AM_SF_ABLATE", 9)
        got = _run(a, am.IGNORE_CASE, b)
        out = (C.c_uint64 * 24)()
        fn = am.libam().am_debug_sfx_roles
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
        am.api.check(fn(out))
        am.debug_set("AM_SF_ABLATE", -1)
        assert got.tobytes() == ref.tobytes()
        r = [int(x) for x in out]
        assert r[3] == (n_bytes + 1023) // 1024               # F: chunks
        assert r[4] == r[13] > 0                              # candidates pushed by the Fs == popped by the Ps
        assert r[14] == r[19] > 0                             # deferred by the Ps == resolved by the Rs
        assert r[20] == len(ref)                              # found == records
    finally:
        am.libam().am_batch_destroy(b)
