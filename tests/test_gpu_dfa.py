"""k_dfa (csrc/am_dfa.hip), the table-walk kernel for dictionaries that meet match-dense text: parity with the oracle and with the suffix-filter
kernel through every entry point that can take the route (am_run / am_count / am_contains_any and their batch forms).  Needs an MI355X.

The flattener gives an image a DFA section by itself only when the automaton is dictionary-like (heavy suffix nodes); the fragment-pool tests force one
(AM_DFA=1) and force the route (am_automaton_set_kernel(a, 3)), so that the kernel meets Unicode case variants, ragged batches and empty haystacks."""
import ctypes as C
import random

import numpy as np
import pytest

import alfred_margaret_amd as am
from alfred_margaret_amd import synth
from oracle import oracle
from tests.helpers import expand_records, fragment_case, oracle_triples

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dfa_everywhere():
    am.debug_set("AM_DFA", 1)
    yield
    am.debug_set("AM_DFA", -1)
    am.debug_set("AM_DFA_CHUNK", -1)
    am.debug_set("AM_DFA_RARE_PERMILLE", -1)


@pytest.fixture()
def dfa_from_one_mib():
    """The table walk for batches of 1 MiB and more (the library's own threshold is 32 MiB: below, a unit's walk costs more than the filter's whole scan)."""
    am.debug_set("AM_DFA_MIN_KIB", 1024)
    yield
    am.debug_set("AM_DFA_MIN_KIB", -1)


def triples(a, case, hays):
    hay, pos, val = a.run_batch_with_case(case, hays)
    return [(int(h), int(p), int(v)) for h, p, v in zip(hay, pos, val)]


def contains_any(a, case, hays):
    s = am.api._Slices(hays)
    out = np.zeros(max(s.n, 1), np.uint8)
    am.api.check(am.api.libam().am_contains_any(a.device, case, s.arr, s.n, out.ctypes.data))
    return [bool(x) for x in out[:s.n]]


def check_dfa_route(needles, hays, case):
    o = oracle.Machine(needles)
    exp = oracle_triples(o, case, hays)
    a = am.Automaton(needles)
    a.set_kernel(3)
    assert triples(a, case, hays) == exp, (case, needles, hays)
    assert [int(c) for c in a.count_matches(case, hays)] == [o.count_matches(case, h) for h in hays], (case, needles, hays)
    assert contains_any(a, case, hays) == [o.contains_any(case, h) for h in hays], (case, needles, hays)
    recs = a.run_records(case, hays)
    assert expand_records(o.values_off(), o.values(), recs["haystack"], recs["state"], recs["end_pos"]) == exp
    keys = [(int(r["haystack"]), int(r["end_pos"])) for r in recs]
    assert keys == sorted(set(keys))                      # one record per position, in (haystack, end_pos) order: no sort behind the two passes


@pytest.mark.parametrize("seed", range(6))
def test_fragment_pool_on_the_table_walk(dfa_everywhere, seed):
    """Even seeds: every byte of the needles has a column.  Odd seeds: four edges in ten may go without (AM_DFA_RARE_PERMILLE), so most steps take the
    rare-byte walk: the state's own edge from the hash, else the question again at its fallback."""
    if seed % 2:
        am.debug_set("AM_DFA_RARE_PERMILLE", 400)
    rng = random.Random(4100 + seed)
    for _ in range(12):
        needles, hays = fragment_case(rng)
        if "" in needles or not any(needles):
            continue                                       # the empty needle: no DFA section (the dense route reports those)
        for case in (0, 1):
            ns = [oracle.lower_utf8(n).decode() for n in needles] if (case and rng.random() < 0.8) else needles
            check_dfa_route(ns, hays, case)


@pytest.mark.parametrize("tune", [3, 1, 0x1000000, 0x103, 0x10, 0x13])
def test_the_walks_variants_report_the_same_records(dfa_everywhere, tune):
    """AM_DFA_TUNE (am_dfa.hip dfa_tune): lanes out of step inside a 16-byte block (3), 16 bytes of text per request (1), no records in LDS (bit 24), no rows in LDS
    with lanes out of step (0x103), one workgroup per CU with all of its LDS (0x10, 0x13: what a device that runs 16 wavefronts per CU at a time gets) -- measurement switches, each the same walk by other loads: records, counts and flags against the oracle, image version 17's
    records (one and two entries, leaning on a row state) on every path."""
    am.debug_set("AM_DFA_TUNE", tune)
    try:
        rng = random.Random(9100 + tune)
        for _ in range(10):
            needles, hays = fragment_case(rng, n_hay_max=8, hay_frags=300)
            if "" in needles or not any(needles):
                continue
            for case in (0, 1):
                ns = [oracle.lower_utf8(n).decode() for n in needles] if case else needles
                check_dfa_route(ns, hays, case)
    finally:
        am.debug_set("AM_DFA_TUNE", -1)


@pytest.mark.parametrize("chunk", [64, 256, 131072])
def test_unit_boundaries_inside_matches_and_code_points(dfa_everywhere, chunk):
    """Small units: every haystack is cut many times, inside needles, inside code points, inside the warm-up of the next unit; haystack boundaries
    and empty haystacks fall inside units.  Units beyond 65 536 bytes (a token's fields are 16 bits): records by count -> scan -> emit, two walks."""
    am.debug_set("AM_DFA_CHUNK", chunk)
    rng = random.Random(77 + chunk)
    for _ in range(6):
        needles, hays = fragment_case(rng, n_hay_max=12, hay_frags=400)
        if "" in needles or not any(needles):
            continue
        hays = hays + ["", hays[0][:3] if hays else "", ""]
        for case in (0, 1):
            ns = [oracle.lower_utf8(n).decode() for n in needles] if case else needles
            check_dfa_route(ns, hays, case)


def test_no_dfa_section_is_an_error_only_when_forced():
    am.debug_set("AM_DFA", 0)                             # (unset, a small automaton gets a DFA section since round 6; a large set of random needles does not: this makes one without)
    a = am.Automaton(["needle", "hay"])
    assert [int(c) for c in a.count_matches(0, ["hay needle hay"])] == [3]
    a.set_kernel(3)
    with pytest.raises(am.AmError):
        a.count_matches(0, ["hay needle hay"])


def test_small_automata_get_the_table_walk_for_match_dense_batches():
    """Round 6: an automaton of up to 32k states carries a DFA section unasked (its table is at most 8 MiB), and a batch of 32 MiB and more is routed by the sample walk: three
    needles that end at EVERY position (a, aa, aaa over a...a: a record per byte, 16 output bytes per input byte) take k_dfa -- tokens for every byte of every unit, the pool's first
    guess from the sample's density --, the same needles over text without them stay on the filter; both equal the oracle on sampled haystacks and the filter on all."""
    import torch
    a = am.Automaton(["a", "aa", "aaa"])
    o = oracle.Machine(["a", "aa", "aaa"])
    dev = torch.device("cuda:0")
    n_hay, hb = 48, 1 << 20
    lib = am.api.libam()
    for fill, want_dfa in ((ord("a"), True), (ord("b"), False)):
        text = torch.full((n_hay * hb + 64,), fill, dtype=torch.uint8, device=dev)
        text[n_hay * hb:] = 0
        offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * hb
        b = C.c_void_p()
        am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_hay * hb, C.byref(b)))
        try:
            res = {}
            for kernel in (0, 2):
                a.set_kernel(kernel)
                am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
                m = C.c_void_p()
                am.api.check(lib.am_run_batch(a.device, 0, b, C.byref(m)))
                torch.cuda.synchronize()
                am.api.check(lib.am_profile_enable(0))
                ms, n = C.c_double(0), C.c_uint64(0)
                lib.am_profile_read(b"dfa", C.byref(ms), C.byref(n))
                if kernel == 0:
                    assert (n.value > 0) == want_dfa
                rs = am.api.matches_of_haystack(m, n_hay - 1)
                res[kernel] = (int(lib.am_matches_size(m)), rs.copy())
                lib.am_matches_free(m)
            assert res[0][0] == res[2][0] == (n_hay * hb if want_dfa else 0)
            assert np.array_equal(res[0][1], res[2][1])
            if want_dfa:
                pos, val = o.run_list(0, bytes([fill]) * 4096)
                st = res[0][1]["state"][:4096].astype(np.int64)
                vo, vals = a.values_off(), a.values()
                got = [(int(p), int(v)) for p, s_ in zip(res[0][1]["end_pos"][:4096], st) for v in vals[int(vo[s_]):int(vo[s_ + 1])]]
                assert got == list(zip(pos.tolist(), val.tolist()))
        finally:
            a.set_kernel(0)
            lib.am_batch_destroy(b)


def test_dictionary_takes_the_table_walk_by_itself(dfa_from_one_mib):
    """The natural-text workload, reduced: the flattener gives the 100k-word dictionary a DFA section, batches of 1 MiB and more take k_dfa without being
    asked (here from 1 MiB on), and its records are those of the suffix-filter kernel and of the oracle."""
    w = synth.WORKLOADS["natural_100k_10GiB"]
    needles = synth.needles_for("natural_100k_10GiB")
    a = am.Automaton(needles)
    o = oracle.Machine(needles)
    cells = 192
    hays = [bytes(synth.haystacks_host(needles, w["mixed"], 7 + i * cells, cells, natural=True)) for i in range(8)]      # 8 x 192 KiB = 1.5 MiB
    lib = am.api.libam()
    am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
    auto = a.run_records(w["case"], hays)
    am.api.check(lib.am_profile_enable(0))
    ms, n = C.c_double(0), C.c_uint64(0)
    am.api.check(lib.am_profile_read(b"dfa", C.byref(ms), C.byref(n)))
    assert n.value == 1, "records come out of ONE walk (tokens + k_dfa_place)"
    a.set_kernel(2)
    sf = a.run_records(w["case"], hays)
    assert np.array_equal(auto, sf) and len(auto) > 100_000
    exp = oracle_triples(o, w["case"], hays[:2])
    sel = auto[auto["haystack"] < 2]
    assert expand_records(o.values_off(), o.values(), sel["haystack"], sel["state"], sel["end_pos"]) == exp
    # a token pool that is far too small: the walk is repeated with the pool the exact counts ask for, the records are the same
    a.set_kernel(0)
    am.debug_set("AM_SF_POOL_BLOCKS", 1)
    try:
        am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
        again = a.run_records(w["case"], hays)
        am.api.check(lib.am_profile_enable(0))
    finally:
        am.debug_set("AM_SF_POOL_BLOCKS", -1)
    am.api.check(lib.am_profile_read(b"dfa", C.byref(ms), C.byref(n)))
    assert n.value == 2 and np.array_equal(again, sf)
    # a token pool the device cannot hold (a batch of hundreds of GiB would ask for one): count -> scan -> emit instead, two walks and no pool
    am.debug_set("AM_SF_POOL_BLOCKS", 1 << 23)
    try:
        am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
        two_pass = a.run_records(w["case"], hays)
        am.api.check(lib.am_profile_enable(0))
    finally:
        am.debug_set("AM_SF_POOL_BLOCKS", -1)
    am.api.check(lib.am_profile_read(b"dfa", C.byref(ms), C.byref(n)))
    assert n.value == 2 and np.array_equal(two_pass, sf)
    am.api.check(lib.am_profile_read(b"dfa_place", C.byref(ms), C.byref(n)))
    assert n.value == 0
    assert [int(c) for c in a.count_matches(w["case"], hays)] == [o.count_matches(w["case"], h) for h in hays]
    # the serialised image carries the DFA section: an automaton attached to it (another process, another GPU: am_multi_*) walks the same table
    attached = am.ImageAutomaton(a.image_bytes(w["case"]))
    am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
    from_image = attached.run_records(w["case"], hays)
    am.api.check(lib.am_profile_enable(0))
    am.api.check(lib.am_profile_read(b"dfa", C.byref(ms), C.byref(n)))
    assert n.value == 1 and np.array_equal(from_image, sf)
    # below the threshold the suffix-filter route stays
    am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
    small = a.run_records(w["case"], hays[:2])
    am.api.check(lib.am_profile_enable(0))
    am.api.check(lib.am_profile_read(b"dfa", C.byref(ms), C.byref(n)))
    assert n.value == 0 and np.array_equal(small, sf[sf["haystack"] < 2])


def test_large_batches_choose_their_route_by_a_sample_walk():
    """From 64 MiB on a dictionary's batch is asked which route it wants (am_abi.cpp make_plan: 4 096 lanes walk 128 bytes each): natural-language text takes the
    table walk, the same dictionary over text in which its words are rare takes the suffix filter; the records are the same either way."""
    import torch
    w = synth.WORKLOADS["natural_100k_10GiB"]
    needles = synth.needles_for("natural_100k_10GiB")
    a = am.Automaton(needles)
    lib = am.api.libam()
    dev = torch.device("cuda:0")
    n_hay, cells = 96, 1024                                                  # 96 MiB
    for natural, expect_dfa in ((True, True), (False, False)):
        text, n_bytes = synth.haystacks_device(needles, w["mixed"], 0, n_hay * cells, dev, natural=natural)
        offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * (cells * 1024)
        b = C.c_void_p()
        am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(b)))
        try:
            counts = {}
            for k in (0, 2, 3):
                a.set_kernel(k)
                am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
                c = np.zeros(n_hay, np.uint64)
                am.api.check(lib.am_count_batch(a.device, w["case"], b, c.ctypes.data, None))
                am.api.check(lib.am_profile_enable(0))
                ms, n = C.c_double(0), C.c_uint64(0)
                am.api.check(lib.am_profile_read(b"dfa", C.byref(ms), C.byref(n)))
                counts[k] = c
                if k == 0:
                    assert (n.value == 1) == expect_dfa, ("natural" if natural else "random", n.value)
            assert np.array_equal(counts[0], counts[2]) and np.array_equal(counts[0], counts[3]) and counts[0].sum() > 0
        finally:
            a.set_kernel(0)
            lib.am_batch_destroy(b)


def test_replacer_over_a_dictionary_scans_with_the_table_walk(dfa_from_one_mib):
    """A Replacer whose needles are a dictionary (the flattener gives its automaton a DFA section): the first scan of a batch of 1 MiB and more takes k_dfa, the
    re-scans of the later passes the suffix filter; the rewritten texts are the oracle's Replacer.run (Replacer.hs:203-274)."""
    from concurrent.futures import ThreadPoolExecutor
    w = synth.WORKLOADS["natural_100k_10GiB"]
    needles = synth.needles_for("natural_100k_10GiB")
    words = [n for n in needles if " " not in n][:30000]
    pairs = [(wd, wd[::-1].upper() if i % 3 else "") for i, wd in enumerate(words)]                  # reversed and upper-cased: replacements rarely make new needles
    cells = 96
    hays = [bytes(synth.haystacks_host(needles, w["mixed"], 11 + i * cells, cells, natural=True)) for i in range(16)] + [b""]      # 16 x 96 KiB = 1.5 MiB
    r = am.Replacer(w["case"], pairs)
    lib = am.api.libam()
    am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
    got = r.run_batch(hays)
    am.api.check(lib.am_profile_enable(0))
    ms, n = C.c_double(0), C.c_uint64(0)
    am.api.check(lib.am_profile_read(b"dfa", C.byref(ms), C.byref(n)))
    assert n.value >= 1, "the first scan of the batch takes the table walk"
    orc = oracle.Replacer(w["case"], pairs)
    with ThreadPoolExecutor(8) as pool:
        exp = list(pool.map(orc.run, hays))
    assert got == exp
    assert sum(g != h for g, h in zip(got, hays)) >= 16


def test_ragged_batch_of_tiny_haystacks_on_the_table_walk(dfa_from_one_mib):
    """2 MiB of natural text cut into 120 000 haystacks of 0-40 bytes: a unit of 2 048 bytes spans a hundred haystacks, most steps take the byte-wise path
    and reset at a boundary.  Records, counts and flags of the table walk (the route such a batch takes by itself) == the suffix filter's; the first haystacks == the oracle."""
    w = synth.WORKLOADS["natural_100k_10GiB"]
    needles = synth.needles_for("natural_100k_10GiB")
    text = bytes(synth.haystacks_host(needles, w["mixed"], 3, 2048, natural=True))
    rng = random.Random(5)
    hays, p = [], 0
    while p < len(text):
        n = rng.choice((0, 0, 3, 7, 16, 17, 31, 40))
        hays.append(text[p:p + n]); p += n
    a = am.Automaton(needles)
    lib = am.api.libam()
    am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
    auto = a.run_records(w["case"], hays)
    am.api.check(lib.am_profile_enable(0))
    ms, n = C.c_double(0), C.c_uint64(0)
    am.api.check(lib.am_profile_read(b"dfa", C.byref(ms), C.byref(n)))
    assert n.value == 1
    counts_auto = a.count_matches(w["case"], hays)
    any_auto = contains_any(a, w["case"], hays)
    a.set_kernel(2)
    sf = a.run_records(w["case"], hays)
    assert np.array_equal(auto, sf) and len(auto) > 10_000
    assert np.array_equal(counts_auto, a.count_matches(w["case"], hays))
    assert any_auto == contains_any(a, w["case"], hays)
    o = oracle.Machine(needles)
    k = 2000
    sel = auto[auto["haystack"] < k]
    assert expand_records(o.values_off(), o.values(), sel["haystack"], sel["state"], sel["end_pos"]) == oracle_triples(o, w["case"], hays[:k])


@pytest.mark.parametrize("chunk", [64, 2048])
def test_sparse_matches_over_many_groups_seal_superblocks_by_age(dfa_everywhere, chunk):
    """A token names its group by a 4-bit ordinal counted from its superblock's first group, so a wavefront's superblock is sealed after 16 of its groups however empty it is
    (k_dfa, round 6).  Text with few matches and small units makes every wavefront take dozens of groups per superblock: cfg3's kind of needles over random text, 1 GiB in units
    of 64 bytes = 262 144 groups of 4 KiB over 8 192 wavefronts, 32 groups each (and 96 MiB in units of 2 048 bytes: superblocks that live for a wavefront's whole share), forced onto
    the table walk -- the records, counts and flags must be the suffix filter's, and the oracle's on sampled haystacks.  Ragged haystacks (1 KiB ... 300 KiB, some empty) put haystack
    boundaries into most groups."""
    import torch
    am.debug_set("AM_DFA_CHUNK", chunk)
    needles = [am.lower_utf8(n).decode("utf-8") for n in synth.make_needles(3000, True)]
    a, o = am.Automaton(needles), oracle.Machine(needles)
    dev = torch.device("cuda:0")
    n_cells = (1024 if chunk == 64 else 96) * 1024
    text, n_bytes = synth.haystacks_device(needles, True, 0, n_cells, dev)
    sizes, at, k = [0], 0, 0
    pattern = (1, 300, 0, 7, 64, 0, 0, 129, 2, 33)
    while at < n_cells:
        at = min(n_cells, at + pattern[k % len(pattern)]); sizes.append(at << 10); k += 1
    offs_h = np.asarray(sizes, dtype=np.int64)
    n_hay = len(offs_h) - 1
    offs = torch.from_numpy(offs_h).to(dev)
    lib = am.api.libam()
    b = C.c_void_p()
    am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(b)))
    try:
        got = {}
        for kernel in (3, 2):
            a.set_kernel(kernel)
            m = C.c_void_p()
            am.api.check(lib.am_run_batch(a.device, 1, b, C.byref(m)))
            recs = am.api.matches_to_numpy(m)
            lib.am_matches_free(m)
            counts = np.zeros(n_hay, np.uint64); tot = C.c_uint64(0)
            am.api.check(lib.am_count_batch(a.device, 1, b, counts.ctypes.data, C.byref(tot)))
            flags = np.zeros(n_hay, np.uint8)
            am.api.check(lib.am_contains_any_batch(a.device, 1, b, flags.ctypes.data))
            got[kernel] = (recs, counts, flags, tot.value)
        assert len(got[3][0]) > 100000
        assert np.array_equal(got[3][0], got[2][0]) and np.array_equal(got[3][1], got[2][1]) and np.array_equal(got[3][2], got[2][2]) and got[3][3] == got[2][3]
        recs = got[3][0]
        first = np.searchsorted(recs["haystack"], np.arange(n_hay + 1))
        vo, vals = a.values_off(), a.values()
        for i in list(range(0, n_hay, max(1, n_hay // 12))) + [n_hay - 1]:
            h = text[int(offs_h[i]):int(offs_h[i + 1])].cpu().numpy()
            pos, val = o.run_list(1, h)
            rs = recs[first[i]:first[i + 1]]
            exp = expand_records(vo, vals, rs["haystack"], rs["state"], rs["end_pos"])
            assert exp == [(i, int(p), int(v)) for p, v in zip(pos, val)], (chunk, i)
    finally:
        a.set_kernel(0)
        lib.am_batch_destroy(b)
