"""Parity of the HIP path (through the C ABI) against the oracle on the same inputs.  Needs an MI355X.

Bit-exactness is the bar (integer/byte work): identical (haystack, matchPos, value) fold sequences,
identical counts, identical flags, identical replaced texts."""
import ctypes as C
import random

import numpy as np
import pytest

import alfred_margaret_amd as am
from alfred_margaret_amd import synth
from oracle import naive, oracle
from tests.conftest import CASES
from tests.helpers import expand_records, fragment_case, oracle_triples

pytestmark = pytest.mark.gpu

KERNELS = {"auto": 0, "ac": 1, "sf": 2}


def product_triples(a, case, hays):
    hay, pos, val = a.run_batch_with_case(case, hays)
    return [(int(h), int(p), int(v)) for h, p, v in zip(hay, pos, val)]


def check_all_paths(needles, hays, case):
    """Every kernel route x every entry point against the oracle."""
    o = oracle.Machine(needles)
    exp = oracle_triples(o, case, hays)
    exp_counts = [o.count_matches(case, h) for h in hays]
    exp_any = [o.contains_any(case, h) for h in hays]
    a = am.Automaton(needles)
    for name, k in KERNELS.items():          # automata with the empty needle too: suffix filter + dense pass (am_dense.hip)
        a.set_kernel(k)
        assert product_triples(a, case, hays) == exp, (name, case, needles, hays)
        assert [int(c) for c in a.count_matches(case, hays)] == exp_counts, (name, case, needles, hays)
        recs = a.run_records(case, hays)
        assert expand_records(o.values_off(), o.values(), recs["haystack"], recs["state"], recs["end_pos"]) == exp
        # records are sorted by (haystack, end_pos), one per position
        keys = [(int(r["haystack"]), int(r["end_pos"])) for r in recs]
        assert keys == sorted(set(keys))
    s = am.Searcher(case, needles)
    assert [bool(x) for x in s.contains_any_batch(hays)] == exp_any


def test_device_is_mi355x():
    info = am.device_info()
    assert info["arch"].startswith("gfx950"), info
    assert info["n_cu"] >= 200


def test_golden_counts(golden):
    for row in golden["count_matches"]:
        if not row["needles"]:
            continue
        a = am.Automaton(row["needles"])
        for k in (1, 2):
            a.set_kernel(k)
            assert int(a.count_matches(CASES[row["case"]], [row["haystack"]])[0]) == row["count"], row["src"]


def test_golden_contains_any(golden):
    for row in golden["contains_any"]:
        s = am.Searcher(CASES[row["case"]], row["needles"])
        assert s.contains_any(row["haystack"]) == row["expected"], row["src"]


def test_golden_match_lists(golden):
    for row in golden["match_lists"]:
        a = am.Automaton(row["needles"])
        pos, val = a.run_with_case(CASES[row["case"]], row["haystack"])
        assert [[int(p), row["needles"][int(v)]] for p, v in zip(pos, val)] == row["matches"], row["src"]


def test_golden_replacer(golden):
    for row in golden["replacer"]:
        r = am.Replacer(CASES[row["case"]], row["pairs"])
        assert r.run(row["haystack"]).decode("utf-8") == row["expected"], row["src"]


def test_golden_contains_all_empty_needle(golden):
    for row in golden["contains_all_empty_needle"]:
        s = am.Searcher(CASES[row["case"]], row["needles"])
        assert s.contains_all(row["haystack"]) == row["expected"], row["src"]


def test_empty_needle_quirk_and_duplicates():
    check_all_paths(["", "ab"], ["xabb", "", "ab"], 0)
    check_all_paths(["b", "ab", "b", "xab"], ["xab", "bxabb"], 0)
    check_all_paths([""], ["abc", ""], 0)
    a = am.Automaton([])
    assert [int(c) for c in a.count_matches(0, ["abc"])] == [0]


def test_empty_needle_with_many_prefix_terminals_stays_on_the_suffix_filter():
    """Round 2 refused automata whose prefix terminals outnumbered 4 x needle ends + 4096 (they fell back to the 50-100 x slower
    general kernel); now the suffix structure takes them.  300 needles of 60 code points that all start with 'a', plus the empty
    needle: ~17 000 prefix states whose last code point starts no needle."""
    rng = random.Random(5)
    needles = [""] + ["a" + "".join(rng.choice("bcdefghij") for _ in range(59)) for _ in range(300)]
    text = "".join(rng.choice(needles[1:])[: rng.randint(1, 60)] + rng.choice("xyz a") for _ in range(400))
    a = am.Automaton(needles)
    a.set_kernel(2)                                    # AM_ERR_UNSUPPORTED here would mean the automaton was refused
    assert len(a.run_records(0, [text])) > 1000
    check_all_paths(needles, [text, "", text[:100], "aab"], 0)
    check_all_paths(needles, [text], 1)


def test_edge_shapes():
    check_all_paths(["a", "aa", "aaa"], ["a" * 300, "", "aa", "b" * 100 + "a"], 0)          # dense output, ragged batch
    check_all_paths(["abc"], [], 0)                                                         # empty batch
    check_all_paths(["abc"], ["", "", ""], 0)                                               # only empty haystacks
    check_all_paths(["x" * 200, "y" + "x" * 199], ["x" * 1000 + "y" + "x" * 400], 0)        # needles longer than a wave step
    check_all_paths(["a\x00b", "\x00"], ["a\x00b\x00\x00a"], 0)                             # NUL is a valid code point (Automaton.hs:499-503)
    check_all_paths(["é", "éé", "日本", "𝄞"], ["éééé日本語𝄞𝄞", "日", "𝄞"], 0)                  # 1-4 byte needles: all suffix tiers


def test_large_result_reaches_the_host_in_pieces():
    """A result of more than 8 MiB of records crosses PCIe through the pinned staging area piece by piece (am_matches_data); twice, so
    that the second result lands in the host block kept from the first."""
    n = 1_300_000                                       # 1.3 M records = 20.8 MB = three pieces
    a = am.Automaton(["a"])
    state = int(a.run_records(0, ["a"])[0]["state"])
    for _ in range(2):
        rec = a.run_records(0, ["a" * n, "b" * 10, "aa"])
        assert len(rec) == n + 2
        assert np.array_equal(rec["end_pos"][:n], np.arange(1, n + 1, dtype=np.uint64))
        assert not rec["haystack"][:n].any() and (rec["state"] == state).all()
        assert [int(x) for x in rec["haystack"][n:]] == [2, 2] and [int(x) for x in rec["end_pos"][n:]] == [1, 2]


def test_one_document_run_has_the_records_with_the_count():
    """Documents up to 64 KiB: count and first records come back in one copy (run_records_small); more than 256 records need the second one."""
    for n in (1, 255, 256, 257, 5000):
        check_all_paths(["a", "ab"], ["a" * n], 0)
    check_all_paths(["abc"], ["x" * 65536], 0)           # the largest document of the short path, no match
    check_all_paths(["abc"], ["x" * 65530 + "abcabc"], 0)


def test_record_pool_overflow_retry(monkeypatch):
    # the single-pass emit guesses its record pool; force a 1-block pool so the retry path runs
    am.debug_set("AM_SF_POOL_BLOCKS", 1)
    check_all_paths(["a", "aa", "aaa", "ab"], ["a" * 3000 + "b" + "a" * 500, "ab" * 700, "", "a"], 0)
    am.debug_set("AM_SF_POOL_BLOCKS", -1)
    check_all_paths(["a", "aa", "aaa", "ab"], ["a" * 3000 + "b" + "a" * 500, "ab" * 700, "", "a"], 0)


@pytest.mark.parametrize("seed", range(8))
def test_fragment_pool(seed):
    rng = random.Random(seed)
    for _ in range(12):
        needles, hays = fragment_case(rng)
        for case in (0, 1):
            ns = [oracle.lower_utf8(n).decode() for n in needles] if (case and rng.random() < 0.8) else needles
            check_all_paths(ns, hays, case)


def test_synthetic_vectors_fixture():
    """Product vs the committed vectors (no oracle involved at run time)."""
    import json, os
    from tests.conftest import ROOT
    data = json.load(open(os.path.join(ROOT, "tests", "golden", "synthetic_vectors.json")))
    for c in data["cases"]:
        a = am.Automaton(c["needles"])
        for k in (2, 1, 0):
            a.set_kernel(k)
            assert product_triples(a, c["case"], c["haystacks"]) == [tuple(t) for t in c["triples"]], (k, c["needles"])


def test_slices_with_offsets():
    # TestInstances.hs:26-33 arbitraryOffset: positions are relative to the slice, not the array
    needles = ["tshirt", "shirts"]
    body = "short tshirts".encode()
    buf = b"zzzshirts" + body + b"tshirtzz"
    a = am.Automaton(needles)
    hay, pos, val = a.run_batch_with_case(0, [(buf, 9, len(body)), body])
    assert [(int(h), int(p), int(v)) for h, p, v in zip(hay, pos, val)] == [(0, 12, 0), (0, 13, 1), (1, 12, 0), (1, 13, 1)]


@pytest.mark.parametrize("workload,n_cells,hay_cells", [("cfg2_runText_10k_1GiB", 4096, 64), ("cfg3_runLower_100k_10GiB", 2048, 256), ("natural_100k_10GiB", 1024, 100)])
def test_synthetic_workload_reduced(workload, n_cells, hay_cells):
    """BASELINE configs at a size the oracle finishes in seconds; full record lists compared.  natural: a dictionary of natural-language words over text
    made of them -- a match every five bytes, heavy suffix nodes with five-byte child entries (k_sf's CHILDREN instantiation), the walker queue."""
    w = synth.WORKLOADS[workload]
    needles = synth.needles_for(workload)
    text = synth.haystacks_host(needles, w["mixed"], 0, n_cells, natural=bool(w.get("natural")))
    hays = [text[i * hay_cells * 1024:(i + 1) * hay_cells * 1024] for i in range(n_cells // hay_cells)]
    o = oracle.Machine(needles)
    a = am.Automaton(needles)
    exp = oracle_triples(o, w["case"], hays)
    assert len(exp) > n_cells // 2
    for k in (2, 1):
        a.set_kernel(k)
        assert product_triples(a, w["case"], hays) == exp, (workload, k)
        assert [int(c) for c in a.count_matches(w["case"], hays)] == [o.count_matches(w["case"], h) for h in hays]


def test_device_generator_matches_host_and_device_batch():
    import torch
    w = synth.WORKLOADS["cfg3_runLower_100k_10GiB"]
    needles = synth.needles_for("cfg3_runLower_100k_10GiB")[:5000]
    n_cells, hay_cells = 1024, 128
    host = synth.haystacks_host(needles, True, 7, n_cells)
    dev, n_bytes = synth.haystacks_device(needles, True, 7, n_cells, torch.device("cuda:0"))
    assert np.array_equal(dev[:n_bytes].cpu().numpy(), host)
    # borrow the HBM-resident batch through am_batch_from_device
    n_hay = n_cells // hay_cells
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device="cuda:0") * (hay_cells * 1024)
    lib = am.api.libam()
    a = am.Automaton(needles)
    b, m = C.c_void_p(), C.c_void_p()
    am.api.check(lib.am_batch_from_device(dev.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(b)))
    try:
        am.api.check(lib.am_run_batch(a.device, w["case"], b, C.byref(m)))
        recs = am.api.matches_to_numpy(m)
        lib.am_matches_free(m)
        counts = np.zeros(n_hay, np.uint64)
        total = C.c_uint64(0)
        am.api.check(lib.am_count_batch(a.device, w["case"], b, counts.ctypes.data, C.byref(total)))
    finally:
        lib.am_batch_destroy(b)
    o = oracle.Machine(needles)
    hays = [host[i * hay_cells * 1024:(i + 1) * hay_cells * 1024] for i in range(n_hay)]
    exp = oracle_triples(o, w["case"], hays)
    assert expand_records(o.values_off(), o.values(), recs["haystack"], recs["state"], recs["end_pos"]) == exp
    assert int(total.value) == len(exp) and [int(c) for c in counts] == [o.count_matches(w["case"], h) for h in hays]


def test_full_size_properties():
    """At a BASELINE-scale size the oracle cannot be run in the test budget; use size-independent
    properties: AC kernel == SF kernel on counts (two independent algorithms), count == number of
    expanded records, every planted needle is found (about one match per 1-KiB cell), sortedness."""
    import torch
    workload = "cfg2_runText_10k_1GiB"
    w = synth.WORKLOADS[workload]
    needles = synth.needles_for(workload)
    n_cells = 256 * 1024                       # 256 MiB
    dev, n_bytes = synth.haystacks_device(needles, w["mixed"], 0, n_cells, torch.device("cuda:0"))
    n_hay = n_cells // 64
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device="cuda:0") * (64 * 1024)
    lib = am.api.libam()
    a = am.Automaton(needles)
    b = C.c_void_p()
    am.api.check(lib.am_batch_from_device(dev.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(b)))
    try:
        totals, per_hay = {}, {}
        for k in (2, 1):
            a.set_kernel(k)
            counts = np.zeros(n_hay, np.uint64)
            total = C.c_uint64(0)
            am.api.check(lib.am_count_batch(a.device, w["case"], b, counts.ctypes.data, C.byref(total)))
            totals[k], per_hay[k] = int(total.value), counts
        assert totals[1] == totals[2] and np.array_equal(per_hay[1], per_hay[2])
        assert totals[2] == int(per_hay[2].sum()) and totals[2] >= n_cells * 0.9
        a.set_kernel(2)
        m = C.c_void_p()
        am.api.check(lib.am_run_batch(a.device, w["case"], b, C.byref(m)))
        recs = am.api.matches_to_numpy(m)
        lib.am_matches_free(m)
        vlen = np.diff(a.values_off())
        assert int(vlen[recs["state"]].sum()) == totals[2]
        key = recs["haystack"].astype(np.uint64) * np.uint64(1 << 32) + recs["end_pos"]
        assert np.all(key[1:] > key[:-1])       # sortedness + one record per position
    finally:
        lib.am_batch_destroy(b)


def test_benchmark_file_protocol(golden):
    """tools/acbench.py: the reference's needles/blank-line/haystack file format and output protocol."""
    import os, subprocess, sys
    from tests.conftest import ROOT
    path = os.path.join(ROOT, "tests", "golden", "benchmark_example.txt")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "acbench.py"), path], capture_output=True, text=True, check=True)
    assert int(r.stderr.strip().splitlines()[-1]) == golden["benchmark_example_file"]["count_naive"]
    assert len(r.stdout.strip().split("\t")) == 5


def test_golden_splitter(golden):
    # AhoCorasickSpec.hs:224-244
    for row in golden["splitter"]:
        sp = am.Splitter(row["sep"])
        got = sp.split_ignore_case(row["haystack"]) if row["ignore_case"] else sp.split(row["haystack"])
        assert [g.decode("utf-8") for g in got] == row["expected"], row["src"]
    # overlapping separators are ignored (Splitter.hs:163-165), empty fragments are kept
    assert am.Splitter("aa").split("aaaXaa") == [b"", b"aX", b""]
    assert am.Splitter(",").split_batch(["a,b,,c", "", ","]) == [[b"a", b"b", b"", b"c"], [b""], [b"", b""]]


def test_replacer_properties():
    rng = random.Random(5)
    for _ in range(30):
        pairs = [("".join(rng.choice("abAB") for _ in range(rng.randint(1, 3))),
                  "".join(rng.choice("abABxyİ") for _ in range(rng.randint(0, 4)))) for _ in range(rng.randint(0, 5))]
        hays = ["".join(rng.choice("abAB" * 10 + "İz") for _ in range(rng.randint(0, 40))) for _ in range(6)]
        got = am.Replacer(0, pairs).run_batch(hays)
        assert [g.decode("utf-8") for g in got] == [naive.sequential_replace(pairs, h) for h in hays]     # AhoCorasickSpec.hs:154-163
        for case in (0, 1):
            assert am.Replacer(case, pairs).run_batch(hays) == [oracle.Replacer(case, pairs).run(h) for h in hays]
    r = am.Replacer(0, [("a", "bbbb")])
    assert r.run_with_limit(8, "aa") == b"bbbbbbbb" and r.run_with_limit(7, "aa") is None


def _replacer_three_ways(case, pairs, hays, max_len=-1):
    """device passes (am_replacer_*) == scans on GPU + host splice == oracle (Replacer.hs:203-242)."""
    r = am.Replacer(case, pairs)
    dev = r.run_batch(hays, max_len)
    dev_stats = r.last_stats()
    host = r.run_batch(hays, max_len, host_splice=True)
    assert r.last_stats()[0] == dev_stats[0]                 # same number of passes
    assert dev_stats[1] <= r.last_stats()[1]                 # the device re-scans only windows around the replacements
    o = oracle.Replacer(case, pairs)
    exp = [o.run(h, max_len) for h in hays]
    assert dev == exp, (case, pairs[:6], max_len)
    assert host == exp
    return dev


def test_replacer_device_overlaps_and_long_match_lists():
    # self-overlapping needles: removeOverlap (Replacer.hs:191-198) across many 64-record batches of one haystack
    _replacer_three_ways(0, [("aa", "b")], ["a" * n for n in (0, 1, 2, 3, 64, 65, 127, 128, 129, 1000, 4097)])
    _replacer_three_ways(0, [("abab", "X"), ("ab", "yy")], ["ab" * 300, "abab" * 77 + "a", "b" + "ab" * 129])
    _replacer_three_ways(0, [("aaa", ""), ("a", "bbbbb")], ["a" * 500, "a" * 7 + "c" + "a" * 200])
    # replacements of every length relation, chains through lower priorities (README.md:67-77 style)
    _replacer_three_ways(0, [("a", "b"), ("b", "c"), ("c", "dd"), ("dd", "")], ["abcabc" * 50, "", "dddd", "x"])
    _replacer_three_ways(0, [("b", "a"), ("a", "b")], ["abba" * 40])
    # tiles: outputs longer than one 16 KiB splice tile, replacement straddling tile borders
    big = ("x" * 1021 + "needle") * 40
    _replacer_three_ways(0, [("needle", "R" * 37), ("xR", "<>")], [big, big[3:], big[:16384], big[:16385]])
    _replacer_three_ways(0, [("x" * 16, "y")], ["x" * 40000])


def test_replacer_device_ignore_case_length_changing():
    # lower-casing that changes byte length inside matches (İ 2→1, ẞ 3→2, K 3→1, Å 3→2): makeMatch (:268-274)
    pairs = [("i", "<I>"), ("ß", "ss"), ("k", "K!"), ("å", "")]
    hays = ["İxİİ", "ẞßẞ", "KkK", "ÅåÅ" * 30, "İẞKÅ" * 100, "aİ" * 70]
    _replacer_three_ways(1, pairs, hays)
    rng = random.Random(11)
    for _ in range(10):
        pairs = [("".join(rng.choice("abikßå") for _ in range(rng.randint(1, 3))),
                  "".join(rng.choice("xyİK") for _ in range(rng.randint(0, 3)))) for _ in range(rng.randint(1, 6))]
        hays = ["".join(rng.choice("abikABIK" * 4 + "İẞKÅßå") for _ in range(rng.randint(0, 300))) for _ in range(9)]
        _replacer_three_ways(1, pairs, hays)
        _replacer_three_ways(0, pairs, hays)


def test_replacer_device_limit_and_empty_needle():
    r = [("a", "bbbb"), ("c", "")]
    hays = ["aa", "a", "", "acac", "cccc", "aaaa" * 10]
    for lim in (0, 1, 4, 7, 8, 9, 40, 160, 161):
        got = _replacer_three_ways(0, r, hays, lim)
        assert got[2] == b""
    # replacementLength counts matches that removeOverlap drops afterwards (Replacer.hs:240 before :241)
    _replacer_three_ways(0, [("aa", "bbb")], ["aaa"], 4)
    _replacer_three_ways(0, [("aa", "bbb")], ["aaa"], 5)
    # the empty needle (general kernel): fires after every successful goto only
    _replacer_three_ways(0, [("", "-"), ("ab", "c")], ["abab", "", "xaby"])
    with pytest.raises(am.AmError) as e:
        am.Replacer(1, [("", "-")]).run("ab")
    assert e.value.code == am.AM_ERR_UNSUPPORTED


def test_replacer_incremental_rescan_equals_full_scans(monkeypatch):
    """Windows + shifted records (am_replace.hip) give the same result, pass count and final text as re-scanning
    everything in every pass (AM_RP_FULL_SCANS=1), on inputs with replacements close together, at haystack borders,
    deletions, growth, and multi-byte text under IgnoreCase."""
    rng = random.Random(77)
    cases = []
    cases.append((0, [("ab", "X"), ("Xc", "abab"), ("ba", ""), ("aX", "yy")], ["abcabcab" * 50, "ab", "bab", "", "cab" * 200]))
    cases.append((0, [("aaa", "a"), ("a", "bb"), ("bbbb", "c")], ["a" * 1000, "a" * 7, "baab" * 100]))
    cases.append((1, [("straße", "STR"), ("i", "İİ"), ("k", ""), ("å", "K")], ["Straße İstanbul KÅ" * 80, "strasse", "ẞ" * 50 + "straße"]))
    for _ in range(8):
        alpha = rng.choice(["abc ", "abİKß", "xyzXYZ"])
        pairs = [("".join(rng.choice(alpha) for _ in range(rng.randint(1, 5))), "".join(rng.choice(alpha + "Q") for _ in range(rng.randint(0, 6)))) for _ in range(rng.randint(2, 25))]
        hays = ["".join(rng.choice(alpha) for _ in range(rng.choice((0, 3, 50, 800, 5000)))) for _ in range(30)]
        cases.append((rng.randint(0, 1), pairs, hays))
    for case, pairs, hays in cases:
        r = am.Replacer(case, pairs)
        inc = r.run_batch(hays)
        inc_stats = r.last_stats()
        am.debug_set("AM_RP_FULL_SCANS", 1)
        full = r.run_batch(hays)
        full_stats = r.last_stats()
        am.debug_set("AM_RP_FULL_SCANS", -1)
        assert inc == full, (case, pairs[:5])
        assert inc_stats[0] == full_stats[0] and inc_stats[1] <= full_stats[1]
        am.debug_set("AM_RP_PIECES", 1)             # small batches take the splicing loop by default: the piece-table loop on the same inputs
        assert r.run_batch(hays) == inc
        am.debug_set("AM_RP_PIECES", -1)
        o = oracle.Replacer(case, pairs)
        assert inc == [o.run(h) for h in hays]


@pytest.mark.parametrize("switch", ["", "AM_RP_NO_FUSE", "AM_RP_NO_SPIN", "AM_RP_MAT_MAIN", "AM_RP_NO_RANGE_REUSE"])
def test_replacer_piece_table_loop_under_its_switches(switch):
    """The piece-table loop hands record ranges from pass to pass, writes finished texts on a second stream and reads its totals by spinning on
    pinned memory; each of these has an A/B switch that is read once per process: the same mixed batches must equal the oracle either way."""
    import os, subprocess, sys
    from tests.conftest import ROOT
    env = dict(os.environ)
    if switch:
        env[switch] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "measure", "replacer_toggles.py")], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "replacer toggles OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_replacer_device_many_haystacks_many_passes():
    rng = random.Random(23)
    alpha = "abcde "
    pairs = [("".join(rng.choice(alpha) for _ in range(rng.randint(2, 4))), "".join(rng.choice("ABC" + alpha) for _ in range(rng.randint(0, 5))))
             for _ in range(60)]
    hays = ["".join(rng.choice(alpha) for _ in range(rng.choice((0, 1, 5, 100, 700, 3000)))) for _ in range(200)]
    for case in (0, 1):
        r = am.Replacer(case, pairs)
        got = r.run_batch(hays)
        passes, scanned = r.last_stats()
        assert passes > 10 and scanned > sum(len(h) for h in hays)
        o = oracle.Replacer(case, pairs)
        assert got == [o.run(h) for h in hays]
        assert got == r.run_batch(hays, host_splice=True)


@pytest.mark.parametrize("groups", ["2", "3", "7"])
def test_replacer_concurrent_haystack_groups(monkeypatch, groups):
    """Large batches run the pass loop on several host threads / streams over groups of haystacks (am_abi.cpp
    replacer_run_groups; from 4096 haystacks by default, forced here on a small ragged batch): same texts, same
    Nothing entries under a length limit, as the oracle's Replacer one haystack at a time (Replacer.hs:203-242)."""
    rng = random.Random(int(groups) * 31)
    alpha = "abcd "
    pairs = [("".join(rng.choice(alpha) for _ in range(rng.randint(2, 4))), "".join(rng.choice("AB" + alpha) for _ in range(rng.randint(0, 6))))
             for _ in range(40)]
    hays = ["".join(rng.choice(alpha) for _ in range(rng.choice((0, 1, 7, 15, 16, 17, 100, 900, 4000)))) for _ in range(157)]
    am.debug_set("AM_RP_GROUPS", int(groups))
    for case in (0, 1):
        r, o = am.Replacer(case, pairs), oracle.Replacer(case, pairs)
        assert r.run_batch(hays) == [o.run(h) for h in hays]
        limited = r.run_batch(hays, max_len=1000)
        assert limited == [o.run(h, 1000) for h in hays]
        assert any(x is None for x in limited) and any(x is not None for x in limited)


def test_replacer_results_left_on_the_device():
    """am_replacer_run_batch_device: same texts and Nothing entries as am_replacer_run_batch, read back one by one with
    am_replaced_read; am_replaced_get hands out device pointers (hipPointerGetAttributes says so)."""
    import torch
    lib = am.api.libam()
    rng = random.Random(5)
    alpha = "abcd "
    pairs = [("".join(rng.choice(alpha) for _ in range(rng.randint(2, 4))), "".join(rng.choice("AB" + alpha) for _ in range(rng.randint(0, 6)))) for _ in range(30)]
    hays = ["".join(rng.choice(alpha) for _ in range(rng.choice((0, 1, 9, 100, 900, 5000)))) for _ in range(120)]
    enc = [h.encode() for h in hays]
    blob = b"".join(enc)
    offs = np.cumsum([0] + [len(e) for e in enc]).astype(np.int64)
    text = torch.frombuffer(bytearray(blob + b"\0" * 16), dtype=torch.uint8).cuda()
    d_offs = torch.from_numpy(offs).cuda()
    for case in (0, 1):
        r, o = am.Replacer(case, pairs), oracle.Replacer(case, pairs)
        batch = C.c_void_p()
        am.api.check(lib.am_batch_from_device(text.data_ptr(), d_offs.data_ptr(), len(hays), len(blob), C.byref(batch)))
        try:
            for limit in (2**64 - 1, 800):
                exp = [o.run(h, -1 if limit > 2**60 else limit) for h in hays]
                for fn, on_dev in ((lib.am_replacer_run_batch, False), (lib.am_replacer_run_batch_device, True)):
                    res = C.c_void_p()
                    am.api.check(fn(C.c_void_p(r.device), batch, C.c_uint64(limit), C.byref(res)))
                    try:
                        assert (lib.am_replaced_device(res) >= 0) == on_dev
                        got = []
                        for i in range(len(hays)):
                            buf, n = C.create_string_buffer(1 << 16), C.c_size_t(0)
                            just = lib.am_replaced_read(res, i, buf, len(buf), C.byref(n))
                            assert just in (0, 1)
                            got.append(buf.raw[:n.value] if just else None)
                        assert got == exp
                        if on_dev:
                            i = max(range(len(hays)), key=lambda k: len(exp[k] or b""))
                            p, n = C.c_void_p(), C.c_size_t(0)
                            assert lib.am_replaced_get(res, i, C.byref(p), C.byref(n)) == 1
                            assert n.value == len(exp[i]) and p.value
                            hip = C.CDLL("libamdhip64.so")
                            attr = C.create_string_buffer(256)          # hipPointerAttribute_t: int type first (2 = device memory)
                            assert hip.hipPointerGetAttributes(attr, p) == 0
                            assert int.from_bytes(attr.raw[:4], "little") == 2
                    finally:
                        lib.am_replaced_free(res)
        finally:
            lib.am_batch_destroy(batch)


def test_single_haystack_split_like_multi_gpu():
    """one big haystack cut into per-rank ranges with a one-match overlap; the union of the ranks'
    own records is the whole-haystack result (here the 'ranks' run one after the other on one GPU)."""
    from alfred_margaret_amd import dist as amdist
    for workload, case in (("cfg2_runText_10k_1GiB", 0), ("cfg3_runLower_100k_10GiB", 1)):
        needles = synth.needles_for(workload)[:5000]
        a = am.Automaton(needles)
        text = bytes(synth.haystacks_host(needles, bool(case), 3, 700))
        whole = a.run_records(case, [text])
        assert len(whole) > 300
        max_cps = max(len(n) if isinstance(n, str) else len(n.decode("utf-8")) for n in needles)
        for world in (2, 5, 8):
            parts = []
            for start, lo, hi, scan_hi in amdist.split_single_haystack(text, world, max_cps):
                parts.append(amdist.own_records(a.run_records(case, [text[start:scan_hi]]), start, lo, hi))
            got = np.concatenate(parts)
            assert np.array_equal(got["end_pos"], whole["end_pos"]) and np.array_equal(got["state"], whole["state"])


def test_contains_all_device_bitmap():
    """Searcher.containsAll (Searcher.hs:173-187): device bitmap fold == host fold of the records == oracle,
    incl. duplicates, the empty needle, no needles (vacuously True) and needle counts around the 32-bit words."""
    rng = random.Random(31)
    for it in range(25):
        n = rng.choice((0, 1, 2, 5, 31, 32, 33, 70))
        needles = ["".join(rng.choice("abcAB") for _ in range(rng.randint(0 if it % 5 == 0 else 1, 3))) for _ in range(n)]
        hays = ["".join(rng.choice("abcABß") for _ in range(rng.choice((0, 3, 40, 400, 3000)))) for _ in range(12)]
        hays.append("".join(needles))           # contains everything (case-sensitively) unless the empty-needle quirk bites
        for case in (0, 1):
            if "" in needles and not any(needles):
                continue
            s = am.Searcher.build_needle_id(case, needles) if hasattr(am.Searcher, "build_needle_id") else am.Searcher(case, needles)
            o = oracle.Machine(needles)
            exp = [o.contains_all(case, h) for h in hays]
            assert [bool(x) for x in s.contains_all_batch(hays)] == exp, (case, needles)      # the direct route: k_sf's ids mode, no record written
            am.debug_set("AM_NO_IDS_SCAN", 1)
            assert [bool(x) for x in s.contains_all_batch(hays)] == exp, (case, needles)      # the record route (k_idset over a full scan)
            am.debug_set("AM_NO_IDS_SCAN", -1)
            assert [bool(x) for x in s.contains_all_batch(hays, host_fold=True)] == exp
    # AhoCorasickSpec.hs:202-218: a haystack made of all needles contains all of them
    needles = synth.needles_for("cfg2_runText_10k_1GiB")[:3000]
    s = am.Searcher(0, needles)
    joined = " ".join(n if isinstance(n, str) else n.decode() for n in needles)
    assert list(s.contains_all_batch([joined, joined[: len(joined) // 2], ""])) == [True, False, False]


def test_contains_all_writes_no_records_and_stops_when_the_set_is_complete():
    """VERDICT r4 item 9 / Searcher.hs:173-187 (`Done` when the set is empty, :181): the default route runs k_sf in its ids mode -- no permute, no
    record fold -- and a document whose first part already contains every needle is not scanned to its end: like containsAny's first match, the full
    row raises the haystack's flag and the wavefronts skip what is left of it.  1-GiB documents resident in HBM (x's around the needles)."""
    import torch
    needles = synth.needles_for("cfg2_runText_10k_1GiB")[:200]
    a = am.Automaton(needles)
    voff = np.ascontiguousarray(a.values_off(), dtype=np.uint64)
    vals = np.ascontiguousarray(a.values(), dtype=np.uint32)
    lib = am.libam()
    dev = torch.device("cuda:0")
    gib = 1 << 30
    joined = torch.tensor(list((" ".join(needles)).encode()), dtype=torch.uint8, device=dev)
    text = torch.full((gib + 64,), ord("x"), dtype=torch.uint8, device=dev)
    offs = torch.tensor([0, gib], dtype=torch.int64, device=dev)
    ids, b = C.c_void_p(), C.c_void_p()
    am.check(lib.am_needle_ids_create(a.device, voff.ctypes.data, vals.ctypes.data, len(needles), C.byref(ids)))
    am.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), 1, gib, C.byref(b)))
    flags = np.zeros(4, np.uint8)

    def run():
        am.check(lib.am_profile_reset()); am.check(lib.am_profile_enable(1))
        am.check(lib.am_contains_all_batch(ids, 0, b, flags.ctypes.data))
        am.check(lib.am_profile_enable(0))
        t = {}
        for k in (b"sf", b"permute", b"idset"):
            ms, n = C.c_double(0), C.c_uint64(0)
            am.check(lib.am_profile_read(k, C.byref(ms), C.byref(n)))
            t[k.decode()] = (ms.value, int(n.value))
        return bool(flags[0]), t

    try:
        run()                                                                      # (haystack index, workspaces)
        text[4096:4096 + joined.numel()] = joined                                  # every needle within the first KiBs
        got, t_early = run()
        assert got is True and t_early["permute"][1] == 0 and t_early["idset"][1] == 0 and t_early["sf"][1] == 1, t_early
        text[4096:4096 + joined.numel()] = ord("x")
        text[gib - 8192:gib - 8192 + joined.numel()] = joined                      # ... within the last ones: the whole document is scanned
        got, t_late = run()
        assert got is True
        text[gib - 8192:gib - 8192 + 40] = ord("x")                                # one needle short
        got, t_never = run()
        assert got is False
        print("containsAll on 1 GiB: complete in the first KiBs %.3f ms, in the last %.3f ms, never %.3f ms" % (t_early["sf"][0], t_late["sf"][0], t_never["sf"][0]))
        assert t_early["sf"][0] * 3 < t_late["sf"][0], (t_early, t_late, t_never)
        am.debug_set("AM_NO_IDS_SCAN", 1)                                          # the record route on the same documents
        got, t_rec = run()
        assert got is False and t_rec["idset"][1] >= 1
        text[gib - 8192:gib - 8192 + joined.numel()] = joined
        got, _ = run()
        am.debug_set("AM_NO_IDS_SCAN", -1)
        assert got is True
    finally:
        lib.am_batch_destroy(b)
        lib.am_needle_ids_destroy(ids)


def test_serialised_image_round_trip(tmp_path):
    """Automaton.save_image -> ImageAutomaton.load: same records without build or flatten."""
    for workload, case in (("cfg2_runText_10k_1GiB", 0), ("cfg3_runLower_100k_10GiB", 1)):
        needles = synth.needles_for(workload)[:4000] + (["", "x"] if case == 0 else [])
        a = am.Automaton(needles)
        hays = [bytes(synth.haystacks_host(needles[:4000], bool(case), 11, 64)), b"", b"xx"]
        path = str(tmp_path / ("image%d.bin" % case))
        a.save_image(path, case)
        b = am.ImageAutomaton.load(path)
        ra, rb = a.run_records(case, hays), b.run_records(case, hays)
        assert len(ra) > 50 and np.array_equal(ra, rb)
        assert np.array_equal(a.count_matches(case, hays), b.count_matches(case, hays))
        with pytest.raises(am.AmError) as e:
            b.run_records(1 - case, hays)          # an image serves its own case mode only
        assert e.value.code == am.AM_ERR_UNSUPPORTED


def _soak_case(rng):
    """Mid-sized adversarial automaton + ragged batch: long needles sharing prefixes/suffixes, needles that are
    substrings of others, every UTF-8 length, case variants, planted needles and near misses."""
    letters = "abcdeABCDE0 " + "äÄßẞéÉσΣςяЯİKÅ" + "𝄞💩"
    stems = ["".join(rng.choice(letters) for _ in range(rng.randint(1, 12))) for _ in range(40)]
    needles = set()
    while len(needles) < 1500:
        n = "".join(rng.choice(stems) for _ in range(rng.randint(1, 4)))
        cut = rng.random()
        if cut < 0.3:
            n = n[rng.randint(0, len(n) - 1):]            # proper suffixes of other needles
        elif cut < 0.5:
            n = n[:rng.randint(1, len(n))]                # prefixes
        if n:
            needles.add(n[:60])
    needles = sorted(needles)
    rng.shuffle(needles)
    needles += [needles[3], needles[7]]                   # duplicates: value order
    hays = []
    for _ in range(rng.randint(20, 40)):
        parts, size = [], rng.choice((0, 1, 7, 300, 5000, 60000))
        while sum(len(p) for p in parts) < size:
            r = rng.random()
            if r < 0.35:
                parts.append(rng.choice(needles))
            elif r < 0.55:
                n = rng.choice(needles)
                i = rng.randrange(len(n))
                parts.append(n[:i] + rng.choice(letters) + n[i + 1:])       # near miss
            elif r < 0.75:
                parts.append(rng.choice(needles).swapcase())
            else:
                parts.append("".join(rng.choice(letters) for _ in range(rng.randint(1, 30))))
        hays.append("".join(parts))
    return needles, hays


@pytest.mark.parametrize("seed", range(4))
def test_soak_random_automata(seed):
    rng = random.Random(9000 + seed)
    needles, hays = _soak_case(rng)
    for case in (0, 1):
        ns = [oracle.lower_utf8(n).decode() for n in needles] if case else needles
        o = oracle.Machine(ns)
        a = am.Automaton(ns)
        exp = oracle_triples(o, case, hays)
        assert len(exp) > 1000
        recs = a.run_records(case, hays)
        assert expand_records(o.values_off(), o.values(), recs["haystack"], recs["state"], recs["end_pos"]) == exp
        assert [int(c) for c in a.count_matches(case, hays)] == [o.count_matches(case, h) for h in hays]
        assert [bool(x) for x in am.Searcher(case, ns).contains_any_batch(hays)] == [o.contains_any(case, h) for h in hays]
        a.set_kernel(1)                                       # the general kernel must agree as well
        recs_ac = a.run_records(case, hays)
        assert np.array_equal(recs_ac, recs)


def test_run_priority_single_pass():
    """am_run_priority = prependMatch/makeMatch (Replacer.hs:252-274) for one pass, per-haystack thresholds."""
    rng = random.Random(41)
    for it in range(12):
        case = it % 2
        alphabet = "abAB" if it < 8 else "aikİKß"
        pairs = [("".join(rng.choice(alphabet) for _ in range(rng.randint(1, 3))), "x" * rng.randint(0, 3)) for _ in range(rng.randint(1, 8))]
        hays = ["".join(rng.choice(alphabet * 3 + "z") for _ in range(rng.choice((0, 5, 60, 700)))) for _ in range(10)]
        thresholds = [rng.choice((1, 0, -1, -3, -100)) for _ in hays]
        r = am.Replacer(case, pairs)
        best, ms = r.run_priority(hays, thresholds)
        needles = [oracle.lower_utf8(n).decode() if case else n for n, _ in pairs]
        o = oracle.Machine(needles)
        got = [(int(m["haystack"]), int(m["start"]), int(m["len"]), int(m["payload"])) for m in ms]
        exp, exp_best = [], []
        for i, h in enumerate(hays):
            hb = h.encode("utf-8")
            pos, val = o.run_list(case, h)
            cands = [(int(p), int(v)) for p, v in zip(pos, val) if -int(v) < thresholds[i]]
            if not cands:
                exp_best.append(-2**63)
                continue
            b = max(-v for _, v in cands)
            exp_best.append(b)
            sel = []
            for p, v in cands:
                if -v != b:
                    continue
                orig = pairs[v][0]
                if case == 0:
                    ln = len(orig.encode("utf-8")); st = p - ln
                else:
                    st = oracle.skip_code_points_backwards(hb, p - 1, len(orig) - 1); ln = p - st
                sel.append((i, st, ln, v))
            exp += sorted(sel)
        assert [int(x) for x in best] == exp_best, (case, pairs, thresholds)
        assert got == exp, (case, pairs)


def test_shared_handles_from_several_threads():
    """Handles are immutable after creation: several host threads may share one automaton / replacer (include/am.h)."""
    from concurrent.futures import ThreadPoolExecutor
    needles = synth.needles_for("cfg2_runText_10k_1GiB")[:3000]
    a = am.Automaton(needles)
    s = am.Searcher(1, [n.lower() if isinstance(n, str) else n for n in needles])
    pairs = [(n, "<%d>" % i) for i, n in enumerate(needles[:200])]
    r = am.Replacer(0, pairs)
    texts = [bytes(synth.haystacks_host(needles, False, 64 * i, 64)) for i in range(6)]
    exp_counts = [int(a.count_matches(0, [t])[0]) for t in texts]
    exp_recs = [a.run_records(0, [t]) for t in texts]
    exp_any = [bool(s.contains_any(t)) for t in texts]
    exp_rep = [r.run(t[:8192]) for t in texts]

    def work(i):
        k = i % len(texts)
        assert int(a.count_matches(0, [texts[k]])[0]) == exp_counts[k]
        assert np.array_equal(a.run_records(0, [texts[k]]), exp_recs[k])
        assert bool(s.contains_any(texts[k])) == exp_any[k]
        assert r.run(texts[k][:8192]) == exp_rep[k]
        return True

    with ThreadPoolExecutor(6) as pool:
        assert all(pool.map(work, range(48)))


def test_replacer_many_tiny_haystacks_large_bookkeeping():
    """More than 2^18 haystacks: the per-pass bookkeeping goes through the multi-launch scans (csrc/am_scan.hip) instead of the single-launch
    k_scan_jobs; same answers."""
    pairs = [("ab", "X"), ("Xc", "ba"), ("b", "yy"), ("zz", "")]
    distinct = ["", "a", "ab", "abc", "cab", "abcab", "zzabzz", "bbbb", "Xc", "abab" * 3]
    o = oracle.Replacer(0, pairs)
    exp = {d: o.run(d) for d in distinct}
    n = (1 << 18) + 1500
    hays = [distinct[(i * 7 + i // 11) % len(distinct)] for i in range(n)]
    r = am.Replacer(0, pairs)
    for loop in (0, -1):                              # the pass-by-pass loop (what this test is about), then the default: one workgroup per haystack in k_rp_loop
        am.debug_set("AM_RP_LOOP", loop)
        got = r.run_batch(hays)
        assert len(got) == n
        bad = [i for i in range(n) if got[i] != exp[hays[i]]]
        assert not bad, (loop, bad[:5], [got[i] for i in bad[:5]])


def test_replacer_record_parallel_fold(monkeypatch):
    """The fold of a pass has two implementations: one wavefront per haystack (many documents) and parallel over the
    records (few documents with very many matches, chosen when records/haystack > 2048).  Both are forced here on the
    same inputs, incl. long runs of overlapping matches (the only serial part of the parallel one)."""
    rng = random.Random(5)
    cases = [
        (0, [("aa", "b")], ["a" * n for n in (0, 1, 2, 3, 64, 65, 127, 128, 129, 1000, 4097)]),
        (0, [("abab", "X"), ("ab", "yy")], ["ab" * 300, "abab" * 77 + "a", "b" + "ab" * 129]),
        (0, [("aaa", ""), ("a", "bbbbb")], ["a" * 500, "a" * 7 + "c" + "a" * 200]),
        (0, [("a", "b"), ("b", "c"), ("c", "dd"), ("dd", "")], ["abcabc" * 50, "", "dddd", "x"]),
        (1, [("i", "<I>"), ("ß", "ss"), ("k", "K!"), ("å", "")], ["İxİİ", "ẞßẞ", "KkK", "ÅåÅ" * 30, "İẞKÅ" * 100, "aİ" * 70]),
        (0, [("a", "bbbb"), ("c", "")], ["aa", "a", "", "acac", "cccc", "aaaa" * 10]),
    ]
    for _ in range(6):
        alpha = rng.choice(["abc ", "abİKß", "xyzXYZ"])
        pairs = [("".join(rng.choice(alpha) for _ in range(rng.randint(1, 4))), "".join(rng.choice(alpha + "Q") for _ in range(rng.randint(0, 5)))) for _ in range(rng.randint(2, 20))]
        hays = ["".join(rng.choice(alpha) for _ in range(rng.choice((0, 3, 50, 800, 6000)))) for _ in range(20)]
        cases.append((rng.randint(0, 1), pairs, hays))
    for case, pairs, hays in cases:
        o = oracle.Replacer(case, pairs)
        for max_len in (-1, 40):
            exp = [o.run(h, max_len) for h in hays]
            r = am.Replacer(case, pairs)
            for forced in ("1", "0"):
                am.debug_set("AM_RP_PARALLEL_FOLD", int(forced))
                assert r.run_batch(hays, max_len) == exp, (forced, case, pairs[:4], max_len)
                am.debug_set("AM_RP_PIECES", 1)     # and with the texts kept as piece tables (the default only for batches of >= 64 documents)
                assert r.run_batch(hays, max_len) == exp, (forced, "pieces", case, pairs[:4], max_len)
                am.debug_set("AM_RP_PIECES", -1)
            am.debug_set("AM_RP_PARALLEL_FOLD", -1)
    # natural trigger: one document with > 2048 matches per pass
    big = ("short tshirts and sweatshirts " * 4000)
    pairs = [("tshirt", "T"), ("shirts", "S"), ("short", "long"), ("and", "&")]
    assert am.Replacer(0, pairs).run(big) == oracle.Replacer(0, pairs).run(big)
    # a periodic document is ONE run of overlapping matches: 1 MB of "a" against "aa" (a million matches, half of them kept);
    # the run is walked by a wavefront, 64 matches per step -- a single thread took about a second for it
    import time
    periodic = "a" * (1 << 20) + "b" + "a" * 4097
    r = am.Replacer(0, [("aa", "c")])
    r.run("aaaa")                                   # warm-up: image upload, workspaces
    t0 = time.perf_counter()
    got = r.run(periodic)
    dt = time.perf_counter() - t0
    assert got == b"c" * (1 << 19) + b"b" + b"c" * 2048 + b"a"
    assert dt < 0.6, dt                              # one thread per run took > 1 s


def test_runtime_knobs_user_stream_and_profiling():
    """am_set_stream: launches go to a caller-supplied HIP stream (here a torch stream); am_profile_*: per-kernel HIP-event
    timing used by bench.py's roofline."""
    import torch
    lib = am.api.libam()
    needles = synth.needles_for("cfg2_runText_10k_1GiB")[:2000]
    a = am.Automaton(needles)
    text = bytes(synth.haystacks_host(needles, False, 5, 256))
    exp = a.run_records(0, [text])
    stream = torch.cuda.Stream()
    am.api.check(lib.am_set_stream(C.c_void_p(stream.cuda_stream)))
    try:
        am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
        got = a.run_records(0, [text])
        am.api.check(lib.am_profile_enable(0))
        assert np.array_equal(got, exp)
        ms, n = C.c_double(0), C.c_uint64(0)
        am.api.check(lib.am_profile_read(b"sf", C.byref(ms), C.byref(n)))
        assert n.value == 1 and 0.0 < ms.value < 100.0
        am.api.check(lib.am_profile_read(b"permute", C.byref(ms), C.byref(n)))
        assert n.value == 1
    finally:
        am.api.check(lib.am_set_stream(None))
    assert np.array_equal(a.run_records(0, [text]), exp)


def test_replacer_map_replacement_and_set_case_sensitivity():
    """Replacer.mapReplacement (Replacer.hs:135-141) and setCaseSensitivity (:148-153): derived replacers without a rebuild."""
    pairs = [("tshirt", "top"), ("shorts", "pants"), ("ß", "ss"), ("k", "x")]
    hays = ["Shorts and TSHIRT, shorts and tshirt K ẞ ß k", "", "tshirtshorts" * 20]
    r = am.Replacer(0, pairs)
    up = r.map_replacement(lambda rep: rep.upper() + b"!")
    fresh = [(n, (rep.upper() + "!")) for n, rep in pairs]
    assert up.run_batch(hays) == [oracle.Replacer(0, fresh).run(h) for h in hays]
    assert r.run_batch(hays) == [oracle.Replacer(0, pairs).run(h) for h in hays]        # the original is untouched
    ic = r.set_case_sensitivity(1)                                                      # needles are lower case already
    assert ic.run_batch(hays) == [oracle.Replacer(1, pairs).run(h) for h in hays]
    assert ic.map_replacement(lambda rep: b"").run_batch(hays) == [oracle.Replacer(1, [(n, "") for n, _ in pairs]).run(h) for h in hays]
    assert ic.set_case_sensitivity(0).run_batch(hays) == r.run_batch(hays)


def test_replacer_compose_property():
    """AhoCorasickSpec.hs:137-148: run (compose a b) == run b . run a, on the device; compose of replacers with different case
    sensitivities is Nothing (Replacer.hs:122-123)."""
    rng = random.Random(41)
    for _ in range(25):
        def pairs():
            return [("".join(rng.choice("abAB\u0130k") for _ in range(rng.randint(1, 3))),
                     "".join(rng.choice("abABxy\u212a") for _ in range(rng.randint(0, 4)))) for _ in range(rng.randint(0, 6))]
        p1, p2 = pairs(), pairs()
        hays = ["".join(rng.choice("abAB" * 6 + "\u0130kKz") for _ in range(rng.randint(0, 60))) for _ in range(8)]
        for case in (0, 1):
            r1, r2 = am.Replacer(case, p1), am.Replacer(case, p2)
            r12 = am.Replacer.compose(r1, r2)
            assert r12 is not None
            step1 = r1.run_batch(hays)
            assert r12.run_batch(hays) == r2.run_batch(step1), (case, p1, p2)
            o12 = oracle.Replacer(case, p1 + p2)
            assert r12.run_batch(hays) == [o12.run(h) for h in hays]
        assert am.Replacer.compose(am.Replacer(0, p1), am.Replacer(1, p2)) is None


def test_run_range_partitions_equal_the_whole_scan():
    """am_run_range / am_count_range (one haystack in ranges with a one-match overlap): ranges that partition (0, len] give,
    concatenated, the records of am_run on the whole haystack -- positions relative to the whole document, cuts inside multi-byte code points,
    IgnoreCase with code points that are longer than what they lower to (K -> k), the empty needle (a record at almost every position)."""
    lib = am.libam()
    rng = random.Random(5)
    cases = []
    needles = synth.needles_for("cfg3_runLower_100k_10GiB")[:5000]
    cases.append((needles, bytes(synth.haystacks_host(needles, True, 3, 300)), 1))
    cases.append((["k", "kk", "åk", "ß", "straße"], ("KKk ÅK Straße ẞ " * 3000).encode(), 1))
    cases.append((["a", "ab", "", "bab"], ("abab baba " * 2000).encode(), 0))
    for needles, text, case in cases:
        a = am.Automaton(needles)
        whole = a.run_records(case, [text])
        total = int(a.count_matches(case, [text])[0])
        vlen = np.diff(a.values_off()).astype(np.int64)
        n = len(text)
        buf = C.create_string_buffer(text, n + 1)
        sl = am.api.Slice(C.addressof(buf), 0, n)
        for world in (1, 2, 3, 7):
            cuts = [0] + sorted(rng.randint(0, n) for _ in range(world - 1)) + [n]
            parts, counts = [], 0
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                m = C.c_void_p()
                am.check(lib.am_run_range(a.device, case, C.byref(sl), lo, hi, C.byref(m)))
                parts.append(am.api.matches_to_numpy(m))
                lib.am_matches_free(m)
                c = C.c_uint64(0)
                am.check(lib.am_count_range(a.device, case, C.byref(sl), lo, hi, C.byref(c)))
                counts += int(c.value)
                assert int(c.value) == int(vlen[parts[-1]["state"].astype(np.int64)].sum()), (lo, hi)      # the count-mode route == the records of the same range
                assert all(lo < int(e) <= hi for e in parts[-1]["end_pos"][:50])
            got = np.concatenate(parts)
            assert got.tobytes() == whole.tobytes(), (world, cuts)
            assert counts == total
        m = C.c_void_p()
        assert lib.am_run_range(a.device, case, C.byref(sl), 5, n + 1, C.byref(m)) == am.AM_ERR_INVALID


def test_threads_started_per_call_do_not_leak_pinned_memory():
    """the Replacer's group threads (and am_multi's per-device threads) are fresh std::threads on every call; their page-locked
    staging buffers are parked when a thread ends and taken over by the next new thread, so repeated calls do not grow page-locked memory."""
    fn = am.libam().am_debug_pinned_bytes
    fn.restype, fn.argtypes = C.c_uint64, []
    rng = random.Random(9)
    pairs = [("".join(rng.choice("abcd") for _ in range(3)), "".join(rng.choice("XY") for _ in range(rng.randint(0, 3)))) for _ in range(20)]
    hays = ["".join(rng.choice("abcd ") for _ in range(rng.choice((50, 900, 4000)))) for _ in range(120)]
    am.debug_set("AM_RP_GROUPS", 3)
    r, o = am.Replacer(0, pairs), oracle.Replacer(0, pairs)
    exp = [o.run(h) for h in hays]
    seen = []
    for it in range(8):
        assert r.run_batch(hays) == exp
        seen.append(int(fn()))
    assert seen[-1] == seen[2], seen                     # flat after the first calls (the pool of parked buffers has reached its size)
    assert seen[-1] < (256 << 20)


@pytest.mark.parametrize("segment_kib", [64, 300])
def test_am_run_in_segments_equals_the_call_in_one_piece(segment_kib):
    """am_run on a large host batch goes up in segments of whole haystacks; a segment's records are rebased on the device and travel back while the next segment is
    uploaded and scanned (am_abi.cpp run_segmented, round 6).  AM_RUN_SEGMENTS = k forces the path on a small input, in segments of k KiB: ragged haystacks (empty
    ones, one larger than a segment), against the same call in one piece and against the oracle; the result lives on the host -- am_matches_haystack_range and
    am_matches_copy answer from there, am_matches_device_data is NULL."""
    rng = random.Random(600 + segment_kib)
    needles = synth.make_needles(3000, True)
    needles = [am.lower_utf8(n).decode("utf-8") for n in needles]
    text = synth.haystacks_host(needles, True, 0, 6000)
    sizes = [rng.choice((0, 1, 700, 5000, 40000, 90000)) for _ in range(60)] + [400 << 10]
    hays, at = [], 0
    for sz in sizes:
        hays.append(bytes(text[at:at + sz])); at += sz
    assert at <= len(text)
    a = am.Automaton(needles)
    o = oracle.Machine(needles)
    lib = am.api.libam()
    s = am.api._Slices(hays)
    lib.am_matches_data.restype = C.c_void_p
    lib.am_matches_device_data.restype = C.c_void_p

    def run(segments):
        am.debug_set("AM_RUN_SEGMENTS", segments)
        try:
            m = C.c_void_p()
            am.api.check(lib.am_run(a.device, 1, s.arr, s.n, C.byref(m)))
            n = int(lib.am_matches_size(m))
            p = lib.am_matches_data(m)
            rec = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n * 2,)).copy() if n else np.zeros(0, np.uint64)
            return m, n, rec
        finally:
            am.debug_set("AM_RUN_SEGMENTS", -1)

    m0, n0, rec0 = run(0)
    m1, n1, rec1 = run(segment_kib)
    try:
        assert n0 == n1 and n0 > 1000 and np.array_equal(rec0, rec1)
        assert lib.am_matches_device_data(m0) and not lib.am_matches_device_data(m1)
        # the oracle, haystack by haystack, through the host result's own accessors
        for h in (0, 5, 17, len(hays) - 1):
            first, count = C.c_uint64(0), C.c_uint64(0)
            am.api.check(lib.am_matches_haystack_range(m1, h, C.byref(first), C.byref(count)))
            got = np.zeros(count.value * 2, np.uint64)
            am.api.check(lib.am_matches_copy(m1, first, count, got.ctypes.data_as(C.c_void_p)))
            pos, _ = o.run_list(1, hays[h])
            assert sorted(set(int(x) for x in pos)) == [int(x) for x in got[0::2]], h
            assert all(int(x) & 0xFFFFFFFF == h for x in got[1::2])
    finally:
        lib.am_matches_free(m0); lib.am_matches_free(m1)


def test_release_device_memory_between_calls():
    """am_release_device_memory frees what the library keeps in HBM between calls (the arrays of freed results, the calling thread's one-shot batch); the next call
    allocates afresh and reports the same records."""
    needles = ["tshirt", "shirts", "shorts", "übergrößen"]
    hays = ["short tshirts and shorts " * 4000, "", "Übergrößen übergrößen"]
    a = am.Automaton(needles)
    o = oracle.Machine(needles)
    lib = am.api.libam()
    exp = oracle_triples(o, 1, hays)
    for _ in range(3):
        recs = a.run_records(1, hays)
        assert expand_records(o.values_off(), o.values(), recs["haystack"], recs["state"], recs["end_pos"]) == exp
        am.api.check(lib.am_release_device_memory())
        am.api.check(lib.am_release_host_memory())
