"""Replacer.run with all passes of a haystack in ONE kernel (csrc/am_rploop.hip, Replacer.hs:203-274): forced with AM_RP_LOOP=1 on the
inputs the pass-by-pass loops are tested on, it must give the oracle's texts, the same Nothing entries and the same number of passes."""
import random
from concurrent.futures import ThreadPoolExecutor

import pytest

import alfred_margaret_amd as am
from alfred_margaret_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu
LDS_SEEN = []               # haystacks each one-kernel run finished out of LDS (am_debug_rp_lds_haystacks)


def _loop_equals_oracle(pairs, hays, max_len=-1, case=0):
    r = am.Replacer(case, pairs)
    am.debug_set("AM_RP_LOOP", 0)
    ref = r.run_batch(hays, max_len)
    ref_stats = r.last_stats()
    am.debug_set("AM_RP_LOOP", 1)
    got = r.run_batch(hays, max_len)                        # k_rp_lds (lists in LDS) + k_rp_loop for the haystacks it gives up
    stats = r.last_stats()
    LDS_SEEN.append(int(am.libam().am_debug_rp_lds_haystacks()))
    am.debug_set("AM_RP_LDS", 0)
    got_global = r.run_batch(hays, max_len)                 # k_rp_loop alone (lists in global memory: round 4's kernel)
    stats_global = r.last_stats()
    am.debug_set("AM_RP_LDS", -1)
    am.debug_set("AM_RP_LOOP", -1)
    o = oracle.Replacer(case, pairs)
    exp = [o.run(h, max_len) for h in hays]
    assert got == exp, (case, pairs[:6], max_len)
    assert got_global == exp, (case, pairs[:6], max_len)
    assert ref == exp
    assert stats[0] == ref_stats[0] == stats_global[0], "number of passes"
    return got, stats


def test_loop_overlaps_chains_and_long_match_lists():
    _loop_equals_oracle([("aa", "b")], ["a" * n for n in (0, 1, 2, 3, 64, 65, 127, 128, 129, 1000, 4097)])
    _loop_equals_oracle([("abab", "X"), ("ab", "yy")], ["ab" * 300, "abab" * 77 + "a", "b" + "ab" * 129])
    _loop_equals_oracle([("aaa", ""), ("a", "bbbbb")], ["a" * 500, "a" * 7 + "c" + "a" * 200])
    _loop_equals_oracle([("a", "b"), ("b", "c"), ("c", "dd"), ("dd", "")], ["abcabc" * 50, "", "dddd", "x"])
    _loop_equals_oracle([("b", "a"), ("a", "b")], ["abba" * 40])
    big = ("x" * 1021 + "needle") * 40
    _loop_equals_oracle([("needle", "R" * 37), ("xR", "<>")], [big, big[3:], big[:16384], big[:16385]])
    _loop_equals_oracle([("x" * 16, "y")], ["x" * 40000])
    _loop_equals_oracle([("ab", "X"), ("Xc", "abab"), ("ba", ""), ("aX", "yy")], ["abcabcab" * 50, "ab", "bab", "", "cab" * 200])
    _loop_equals_oracle([("aaa", "a"), ("a", "bb"), ("bbbb", "c")], ["a" * 1000, "a" * 7, "baab" * 100])


def test_loop_length_limit_and_multibyte_text():
    r = [("a", "bbbb"), ("c", "")]
    hays = ["aa", "a", "", "acac", "cccc", "aaaa" * 10]
    for lim in (0, 1, 4, 7, 8, 9, 40, 160, 161):
        got, _ = _loop_equals_oracle(r, hays, lim)
        assert got[2] == b""
    _loop_equals_oracle([("aa", "bbb")], ["aaa"], 4)
    _loop_equals_oracle([("aa", "bbb")], ["aaa"], 5)
    _loop_equals_oracle([("ß", "ss"), ("İ", "i"), ("ss", "ẞ"), ("å", "")], ["ẞßẞ", "ÅåÅ" * 30, "İẞKÅß" * 100, "aİ" * 70, "straße" * 40])


def test_loop_ignore_case_walks_back_through_the_piece_list():
    """makeMatch of an IgnoreCase replacer (Replacer.hs:268-274): the match is as long as its code points are in the HAYSTACK (İ 2 -> 1 bytes,
    ẞ 3 -> 2, K 3 -> 1, Å 3 -> 2 under lower-casing), found by skipCodePointsBackwards -- here through the piece list, after earlier passes have cut
    the text into pieces."""
    pairs = [("i", "<I>"), ("ß", "ss"), ("k", "K!"), ("å", "")]
    hays = ["İxİİ", "ẞßẞ", "KkK", "ÅåÅ" * 30, "İẞKÅ" * 100, "aİ" * 70]
    _loop_equals_oracle(pairs, hays, case=1)
    _loop_equals_oracle([("straße", "STR"), ("i", "İİ"), ("k", ""), ("å", "K")], ["Straße İstanbul KÅ" * 80, "strasse", "ẞ" * 50 + "straße"], case=1)
    rng = random.Random(11)
    for _ in range(10):
        pairs = [("".join(rng.choice("abikßå") for _ in range(rng.randint(1, 3))),
                  "".join(rng.choice("xyİK") for _ in range(rng.randint(0, 3)))) for _ in range(rng.randint(1, 6))]
        hays = ["".join(rng.choice("abikABIK" * 4 + "İẞKÅßå") for _ in range(rng.randint(0, 300))) for _ in range(rng.choice((9, 80)))]
        _loop_equals_oracle(pairs, hays, case=1)
        _loop_equals_oracle(pairs, hays, 400, case=1)


def test_loop_random_pair_sets_many_haystacks_many_passes():
    rng = random.Random(41)
    for _ in range(8):
        alpha = rng.choice(["abc ", "abİKß", "xyzXYZ", "abcde "])
        pairs = [("".join(rng.choice(alpha) for _ in range(rng.randint(1, 5))), "".join(rng.choice(alpha + "Q") for _ in range(rng.randint(0, 6)))) for _ in range(rng.randint(2, 60))]
        pairs = [p for p in pairs if p[0]]
        hays = ["".join(rng.choice(alpha) for _ in range(rng.choice((0, 1, 3, 50, 800, 5000)))) for _ in range(rng.choice((1, 30, 200)))]
        _, stats = _loop_equals_oracle(pairs, hays)
        _loop_equals_oracle(pairs, hays, 1000)
        _loop_equals_oracle(pairs, hays, case=1)


def test_loop_falls_back_when_a_haystack_outgrows_its_regions():
    """Every haystack's record and piece lists are sized from its first scan (2 n + 64 records); replacements that double the number of
    matches per pass outgrow them, the kernel raises its overflow flag and the batch takes the pass-by-pass loop: same answer."""
    pairs = [("a", "bb"), ("b", "cc"), ("c", "dd"), ("d", "ee")]
    hays = ["a" * 200, "abcd" * 64, "x"]
    got, stats = _loop_equals_oracle(pairs, hays)
    assert got[0] == b"e" * 3200 and stats[0] == 4


def test_long_match_lists_take_the_pass_by_pass_loop():
    """A wavefront walks its haystack's whole record list in every pass, so a batch in which some document has more than 4 096 matches goes to the
    pass-by-pass loop (parallel over the records) by default; forced through the one-kernel loop it gives the same texts."""
    pairs = [("aa", "b"), ("bb", "a"), ("ab", "cc")]
    hays = ["abba" * 20] * 70 + ["aa" * 6000, "x"]
    r = am.Replacer(0, pairs)
    got = r.run_batch(hays)
    o = oracle.Replacer(0, pairs)
    assert got == [o.run(h) for h in hays]
    am.debug_set("AM_RP_LOOP", 1)
    assert r.run_batch(hays) == got
    am.debug_set("AM_RP_LOOP", -1)


def test_loop_cfg5_reduced_and_what_it_scans():
    """BASELINE configs[4] reduced to 256 x 64 KiB: the default route for batches of many documents; the oracle on the first 48 of them."""
    workload = "cfg5_replacer_50k_1GiB"
    w = synth.WORKLOADS[workload]
    pairs = synth.replacer_pairs(workload)
    n_hay, hb = 256, w["hay_bytes"]
    host = synth.haystacks_host([p[0] for p in pairs], w["mixed"], 0, n_hay * hb // synth.CELL)
    hays = [bytes(host[i * hb:(i + 1) * hb]) for i in range(n_hay)] + [b"", bytes(host[:100])]
    r = am.Replacer(w["case"], pairs)
    got = r.run_batch(hays)                                  # switches unset: >= 64 documents take the one-kernel loop
    passes, scanned = r.last_stats()
    am.debug_set("AM_RP_LOOP", 0)
    ref = r.run_batch(hays)
    ref_passes, ref_scanned = r.last_stats()
    am.debug_set("AM_RP_LOOP", -1)
    assert got == ref and passes == ref_passes and passes > 100
    total = sum(len(h) for h in hays)
    assert total < scanned <= ref_scanned < total * 4                    # windows as long as the longest needle needs (the pass-by-pass loop: 4 bytes per code point + 4)
    orc = oracle.Replacer(w["case"], pairs)
    with ThreadPoolExecutor(8) as pool:
        exp = list(pool.map(orc.run, hays[:48]))
    assert got[:48] == exp


def test_loop_reach_comes_from_the_automaton_not_from_the_payload_lengths():
    """a replacer built IgnoreCase holds LOWER-CASED needles, and lower-casing can add bytes (U+023A, 2 bytes -> U+2C65, 3 bytes;
    U+023E -> U+2C66) while the payloads keep the ORIGINAL lengths (Replacer.hs:112-113).  setCaseSensitivity CaseSensitive (Replacer.hs:148-153) then runs
    those 3-byte needles CaseSensitive: the re-scan reach of the one-kernel loop must be the automaton's longest needle in bytes, or matches that start more
    than `payload bytes` before a replacement are missed.  AM_RP_LOOP=1 == AM_RP_LOOP=0 == the oracle with the same setCaseSensitivity."""
    needles = ["ȺȾȺȾȺȾȺȾ", "Ⱥb", "ȾȺȾȺȾȺȾa", "xȺ"]
    pairs = list(zip(needles, ["<1>", "Ⱦ", "ⱥⱦⱥⱦⱥⱦⱥⱦ", "ⱦⱥⱦⱥⱦⱥ"]))
    low = "ⱥⱦ"
    rng = random.Random(23)
    hays = ["".join(rng.choice([low[0], low[1], "a", "b", "x", "ⱥⱦ" * 4]) for _ in range(rng.randint(0, 400))) for _ in range(96)]
    hays += ["xⱥⱦⱥⱦⱥⱦⱥⱦⱥⱦa" * 30, "ⱦⱥⱦⱥⱦⱥⱦ" + "a" * 5 + "xⱥb", ""]
    r = am.Replacer(1, pairs).set_case_sensitivity(0)
    o = oracle.Replacer(1, pairs).set_case_sensitivity(0)
    exp = [o.run(h) for h in hays]
    assert any(e != h.encode() for e, h in zip(exp, hays))
    am.debug_set("AM_RP_LOOP", 0)
    ref = r.run_batch(hays)
    am.debug_set("AM_RP_LOOP", 1)
    got = r.run_batch(hays)
    am.debug_set("AM_RP_LOOP", -1)
    assert ref == exp
    assert got == exp
    assert r.run_batch(hays) == exp                          # the default route (>= 64 documents: the one-kernel loop)


def test_the_lds_kernel_is_what_runs():
    """(after the tests above, same process) the runs of this file went through k_rp_lds for most haystacks -- and not for all: the long lists, the
    growing ones and the windows with many new records took k_rp_loop, which the same runs compared with AM_RP_LDS=0 as well."""
    if not LDS_SEEN:
        pytest.skip("runs on its own: nothing recorded")
    assert sum(1 for n in LDS_SEEN if n > 0) >= len(LDS_SEEN) // 2, LDS_SEEN
