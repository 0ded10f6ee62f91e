"""BASELINE.json configs[3] / configs[4] and the at-scale parity gate, on the MI355X through the C ABI.

  cfg4  100k needles, 100-KiB haystacks, the batch cut into contiguous blocks per rank (dist.shard_bounds)
  cfg5  Replacer.run with the full 50k (needle, replacement) pair set
  fold checksum  am_matches_fold_hash == the oracle's runWithCase folded with the same hash function
"""
import ctypes as C
import functools
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import alfred_margaret_amd as am
from alfred_margaret_amd import dist as amdist
from alfred_margaret_amd import synth
from oracle import oracle
from tests.helpers import expand_records, fragment_case, oracle_triples

pytestmark = pytest.mark.gpu


@functools.lru_cache(maxsize=2)
def _cfg3():
    needles = synth.needles_for("cfg3_runLower_100k_10GiB")
    return needles, am.Automaton(needles), oracle.Machine(needles)


def _device_batch(needles, mixed, first_cell, n_hay, hay_bytes):
    import torch
    dev = torch.device("cuda:0")
    text, n_bytes = synth.haystacks_device(needles, mixed, first_cell, n_hay * hay_bytes // synth.CELL, dev)
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * hay_bytes
    return text, offs, n_bytes


def _run_device(a, case, text_ptr, offs_ptr, n_hay, n_bytes, table=None):
    lib = am.api.libam()
    b, m = C.c_void_p(), C.c_void_p()
    am.api.check(lib.am_batch_from_device(text_ptr, offs_ptr, n_hay, n_bytes, C.byref(b)))
    try:
        am.api.check(lib.am_run_batch(a.device, case, b, C.byref(m)))
        recs = am.api.matches_to_numpy(m)
        hashes = table.fold_hash(m, n_hay) if table is not None else None
        lib.am_matches_free(m)
    finally:
        lib.am_batch_destroy(b)
    return recs, hashes


def test_cfg4_shape_sharded_equals_unsharded_equals_oracle():
    """BASELINE configs[3]: the cfg3 automaton (100k needles, IgnoreCase) over 2048 x 100-KiB haystacks; scanning the
    contiguous per-rank blocks of dist.shard_bounds (world 2 and 8) and concatenating with dist.gather_records gives
    the unsharded record array; a spread sample of haystacks is compared record by record with the oracle."""
    import torch
    w = synth.WORKLOADS["cfg4_100k_1M_haystacks"]
    needles, a, o = _cfg3()
    n_hay, hb = 2048, w["hay_bytes"]
    text, offs, n_bytes = _device_batch(needles, w["mixed"], 0, n_hay, hb)
    a.set_kernel(0)
    whole, _ = _run_device(a, w["case"], text.data_ptr(), offs.data_ptr(), n_hay, n_bytes)
    assert len(whole) > n_hay * 90                       # about one planted needle per KiB plus incidental ones
    key = whole["haystack"].astype(np.uint64) * np.uint64(1 << 32) + whole["end_pos"]
    assert np.all(key[1:] > key[:-1])
    for world in (2, 8):
        parts = []
        for r in range(world):
            lo, hi = amdist.shard_bounds(n_hay, r, world)
            o_local = torch.arange(hi - lo + 1, dtype=torch.int64, device=text.device) * hb
            local, _ = _run_device(a, w["case"], text.data_ptr() + lo * hb, o_local.data_ptr(), hi - lo, (hi - lo) * hb)
            parts.append(amdist.gather_records(local, lo))
        got = np.concatenate(parts)
        assert got.tobytes() == whole.tobytes(), world
    # oracle on every 32nd haystack: full (matchPos, value) fold sequences
    host = text[:n_bytes].cpu().numpy()
    sample = list(range(0, n_hay, 32))
    first = np.searchsorted(whole["haystack"], np.arange(n_hay + 1))
    for h in sample:
        pos, val = o.run_list(w["case"], host[h * hb:(h + 1) * hb])
        rs = whole[first[h]:first[h + 1]]
        got = expand_records(o.values_off(), o.values(), rs["haystack"], rs["state"], rs["end_pos"])
        assert got == [(h, int(p), int(v)) for p, v in zip(pos, val)], h


def test_cfg4_general_kernel_agrees_on_a_block():
    """The independent algorithm (general AC walk) on one rank's block of the cfg4 shape."""
    w = synth.WORKLOADS["cfg4_100k_1M_haystacks"]
    needles, a, _ = _cfg3()
    n_hay, hb = 256, w["hay_bytes"]
    text, offs, n_bytes = _device_batch(needles, w["mixed"], 5 * 100, n_hay, hb)
    out = {}
    for k in (2, 1):
        a.set_kernel(k)
        out[k], _ = _run_device(a, w["case"], text.data_ptr(), offs.data_ptr(), n_hay, n_bytes)
    a.set_kernel(0)
    assert out[1].tobytes() == out[2].tobytes() and len(out[2]) > 10000


@pytest.mark.parametrize("full_scans", [False, True])
def test_cfg5_replacer_reduced(monkeypatch, full_scans):
    """BASELINE configs[4]: Replacer.run with the full cfg5 pair set (50 000 pairs, ~160 passes) over 64 x 64 KiB of the
    cfg5 haystack generator: device passes == the oracle's Replacer (Replacer.hs:203-242), with the incremental
    re-scan and with a full scan in every pass (AM_RP_FULL_SCANS=1)."""
    workload = "cfg5_replacer_50k_1GiB"
    w = synth.WORKLOADS[workload]
    pairs = synth.replacer_pairs(workload)
    assert len(pairs) == 50_000
    n_hay, hb = 64, w["hay_bytes"]
    host = synth.haystacks_host([p[0] for p in pairs], w["mixed"], 0, n_hay * hb // synth.CELL)
    hays = [bytes(host[i * hb:(i + 1) * hb]) for i in range(n_hay)] + [b"", bytes(host[:100])]
    if full_scans:
        am.debug_set("AM_RP_FULL_SCANS", 1)
    r = am.Replacer(w["case"], pairs)
    got = r.run_batch(hays)
    passes, scanned = r.last_stats()
    assert passes > 100
    total = sum(len(h) for h in hays)
    assert (scanned > total * 10) if full_scans else (scanned < total * 4)
    orc = oracle.Replacer(w["case"], pairs)
    with ThreadPoolExecutor(8) as pool:                   # ctypes releases the GIL
        exp = list(pool.map(orc.run, hays))
    assert got == exp
    assert sum(g != h for g, h in zip(got, hays)) >= n_hay    # every 64-KiB haystack was rewritten


def test_fold_hash_equals_oracle_fold():
    """am_matches_fold_hash (device) == runWithCase folded with the same hash function in the oracle: random fragment
    automata (both kernels, duplicates, empty needle, empty haystacks) and the cfg3 automaton on 48 x 128 KiB."""
    rng = random.Random(99)
    for _ in range(40):
        needles, hays = fragment_case(rng, n_hay_max=6)
        for case in (0, 1):
            ns = [oracle.lower_utf8(n).decode() for n in needles] if case else needles
            o, a = oracle.Machine(ns), am.Automaton(ns)
            table = am.ValuesTable(a)
            exp = [o.fold_hash(case, h) for h in hays]
            for k in (1, 2):
                a.set_kernel(k)
                s = am.api._Slices(hays)
                m = C.c_void_p()
                am.api.check(am.api.libam().am_run(a.device, case, s.arr, s.n, C.byref(m)))
                hashes, counts = table.fold_hash(m, len(hays))
                am.api.libam().am_matches_free(m)
                assert [(int(h), int(c)) for h, c in zip(hashes, counts)] == exp, (ns, hays, case, k)
    w = synth.WORKLOADS["cfg3_runLower_100k_10GiB"]
    needles, a, o = _cfg3()
    n_hay, hb = 48, 128 << 10
    text, offs, n_bytes = _device_batch(needles, w["mixed"], 11, n_hay, hb)
    host = text[:n_bytes].cpu().numpy()
    table = am.ValuesTable(a)
    exp = [o.fold_hash(w["case"], host[i * hb:(i + 1) * hb]) for i in range(n_hay)]
    for k in (2, 1):
        a.set_kernel(k)
        _, (hashes, counts) = _run_device(a, w["case"], text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, table)
        assert [(int(h), int(c)) for h, c in zip(hashes, counts)] == exp, k
    a.set_kernel(0)


def test_cfg1_contains_any_three_needles_one_megabyte():
    """BASELINE.json configs[0]: Searcher.containsAny CaseSensitive, needles tshirt / shirts / shorts over 1 MB of synthetic ASCII
    (Searcher.hs:156-164), through the one-shot entry points a Haskell caller binds: am_contains_any, am_count and am_run on a host
    slice == the oracle, plus haystack sizes around the light / full kernel configuration switch (16 KiB) and the pinned-upload
    piece size (256 KiB)."""
    needles = ["tshirt", "shirts", "shorts"]
    rng = np.random.default_rng(1)
    words = ["short", "tshirts", "sweatshirts", "and", "shirtshirts", "the", "quick", "brown", "fox", "shorts"]
    big = (" ".join(words[int(i)] for i in rng.integers(0, len(words), size=200000)))[:1_000_000].encode()
    quiet = bytes(rng.integers(97, 123, size=1_000_000, dtype=np.uint8)).replace(b"sh", b"xx")       # 1 MB without any needle
    a = am.Automaton(needles)
    s = am.Searcher(0, needles)
    o = oracle.Machine(needles)
    for hay in (big, quiet, big[:262144], big[:262145], big[:300_001], big[:16384], big[:16385], big[:100], b""):
        assert bool(s.contains_any(hay)) == o.contains_any(0, hay)
        assert int(a.count_matches(0, [hay])[0]) == o.count_matches(0, hay)
    hay_i, pos, val = a.run_batch_with_case(0, [big, quiet])
    p_, v_ = o.run_list(0, big)
    assert [(int(h), int(p), int(v)) for h, p, v in zip(hay_i, pos, val)] == [(0, int(p), int(v)) for p, v in zip(p_, v_)]
    assert len(pos) > 100_000 and o.contains_any(0, quiet) is False


def test_contains_any_stops_at_the_first_match():
    """Searcher.containsAny ends its fold at the first match (Searcher.hs:156-164 `Done True`, Automaton.hs:528-532).  In flag mode k_sf skips a
    chunk whose haystack is already flagged: a 1-GiB document that matches in its first KiB costs a few chunks per wavefront instead of the scan;
    the flags are what they were without the short cut (same haystacks as the oracle's containsAny)."""
    import torch
    needles, a, o = _cfg3()
    dev = torch.device("cuda:0")
    lib = am.api.libam()
    gib = 1 << 30
    text = torch.full((gib + 64,), ord("x"), dtype=torch.uint8, device=dev)       # no needle of the benchmark set is made of x's only
    needle = needles[0].encode()
    text[100:100 + len(needle)] = torch.tensor(list(needle), dtype=torch.uint8, device=dev)
    offs = torch.tensor([0, gib], dtype=torch.int64, device=dev)
    b = C.c_void_p()
    am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), 1, gib, C.byref(b)))
    flags = np.zeros(4, np.uint8)
    try:
        am.api.check(lib.am_contains_any_batch(a.device, am.IGNORE_CASE, b, flags.ctypes.data))      # (builds the haystack index)
        assert flags[0] == 1
        am.api.check(lib.am_profile_enable(1)); am.api.check(lib.am_profile_reset())
        am.api.check(lib.am_contains_any_batch(a.device, am.IGNORE_CASE, b, flags.ctypes.data))
        ms, launches = C.c_double(0), C.c_uint64(0)
        am.api.check(lib.am_profile_read(b"sf", C.byref(ms), C.byref(launches)))
        hit_ms = ms.value / max(launches.value, 1)
        # the same document without the match: the whole scan
        text[100:100 + len(needle)] = ord("x")
        am.api.check(lib.am_profile_reset())
        am.api.check(lib.am_contains_any_batch(a.device, am.IGNORE_CASE, b, flags.ctypes.data))
        am.api.check(lib.am_profile_read(b"sf", C.byref(ms), C.byref(launches)))
        miss_ms = ms.value / max(launches.value, 1)
        am.api.check(lib.am_profile_enable(0))
        assert flags[0] == 0
        print("containsAny on 1 GiB: first-KiB match %.3f ms, no match %.3f ms" % (hit_ms, miss_ms))
        assert hit_ms < 0.15 and hit_ms * 3 < miss_ms
    finally:
        lib.am_batch_destroy(b)
    # several haystacks, some matching early, some late, some not at all: the flags are the oracle's containsAny
    hays = [b"x" * 300_000 + needle + b"x" * 50, needle + b"y" * 400_000, b"z" * 200_000, b"", b"q" * 70_000 + needle.upper()]
    got = am.Searcher(am.IGNORE_CASE, needles[:2000]).contains_any_batch(hays)
    o2 = oracle.Machine(needles[:2000])
    exp = [bool(o2.contains_any(am.IGNORE_CASE, h)) for h in hays]
    assert exp == [True, True, False, False, True]
    assert [bool(g) for g in got] == exp


def test_cfg2_single_haystack_equals_the_oracle():
    """BASELINE configs[1], SURVEY 8d shape (i): ONE large haystack (here 64 MiB of the 1-GiB workload `cfg2_single_1GiB`; the reference scans a
    Text of any size in one fold, Automaton.hs:468-480).  Every (matchPos, value) of the fold against the oracle, and the general kernel on
    the same document."""
    import torch
    w = synth.WORKLOADS["cfg2_single_1GiB"]
    needles = synth.needles_for("cfg2_single_1GiB")
    assert len(needles) == 10_000 and w["n_hay"] == 1
    a, o = am.Automaton(needles), oracle.Machine(needles)
    n_bytes = 64 << 20
    dev = torch.device("cuda:0")
    text, _ = synth.haystacks_device(needles, w["mixed"], 0, n_bytes // synth.CELL, dev)
    offs = torch.tensor([0, n_bytes], dtype=torch.int64, device=dev)
    out = {}
    for k in (2, 1):
        a.set_kernel(k)
        out[k], _ = _run_device(a, w["case"], text.data_ptr(), offs.data_ptr(), 1, n_bytes)
    a.set_kernel(0)
    assert out[1].tobytes() == out[2].tobytes()
    recs = out[2]
    assert len(recs) > 64 * 1024 and not recs["haystack"].any()
    assert np.all(recs["end_pos"][1:] > recs["end_pos"][:-1]) and int(recs["end_pos"][-1]) > (63 << 20)      # positions are 64-bit offsets into the ONE document
    pos, val = o.run_list(w["case"], text[:n_bytes].cpu().numpy())
    got = expand_records(o.values_off(), o.values(), recs["haystack"], recs["state"], recs["end_pos"])
    assert len(got) == len(pos)
    assert got == [(0, int(p), int(v)) for p, v in zip(pos, val)]
