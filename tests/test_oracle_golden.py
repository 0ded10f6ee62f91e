"""Pins oracle/am_oracle.c against the reference's own known-answer tests (tests/golden/) and
against the independent naive oracle.  CPU only."""
import random

import pytest

from oracle import naive, oracle
from tests.conftest import CASES


def test_utf8_encoding(golden):
    for row in golden["utf8_encoding"]:
        assert list(row["text"].encode("utf-8")) == row["bytes"], row["src"]


def test_count_matches(golden):
    for row in golden["count_matches"]:
        if not row["needles"]:
            continue  # the reference short-circuits needles == [] to 0 without building
        m = oracle.Machine(row["needles"])
        assert m.count_matches(CASES[row["case"]], row["haystack"]) == row["count"], row["src"]


def test_contains_any(golden):
    for row in golden["contains_any"]:
        m = oracle.Machine(row["needles"])
        assert m.contains_any(CASES[row["case"]], row["haystack"]) == row["expected"], row["src"]


def test_match_lists(golden):
    for row in golden["match_lists"]:
        m = oracle.Machine(row["needles"])
        pos, val = m.run_list(CASES[row["case"]], row["haystack"])
        got = [[int(p), row["needles"][int(v)]] for p, v in zip(pos, val)]
        assert got == row["matches"], row["src"]


def test_replacer(golden):
    for row in golden["replacer"]:
        r = oracle.Replacer(CASES[row["case"]], row["pairs"])
        assert r.run(row["haystack"]).decode("utf-8") == row["expected"], row["src"]


def test_contains_all_empty_needle_quirk(golden):
    for row in golden["contains_all_empty_needle"]:
        m = oracle.Machine(row["needles"])
        assert m.contains_all(CASES[row["case"]], row["haystack"]) == row["expected"], row["src"]
    # the empty needle IS reported after every successful transition (Automaton.hs:373-376,502-503)
    m = oracle.Machine(["", "ab"])
    pos, val = m.run_list(0, "xabb")
    assert [(int(p), int(v)) for p, v in zip(pos, val)] == [(2, 0), (3, 1), (3, 0)]


def test_skip_code_points_backwards(golden):
    for row in golden["skip_code_points_backwards"]:
        if row["expected"] == "error":
            with pytest.raises(IndexError):
                oracle.skip_code_points_backwards(row["text"], row["index"], row["n"])
        else:
            assert oracle.skip_code_points_backwards(row["text"], row["index"], row["n"]) == row["expected"], row


def test_lower_code_point_vs_unlower_table(golden):
    for row in golden["unlower"]:
        cp = ord(row["cp"])
        for member in row["set"]:
            assert oracle.lower_code_point(ord(member)) == cp, row["src"]
        if not row["set"]:
            assert oracle.lower_code_point(cp) != cp, row["src"]


def test_lower_code_point_all_unicode_matches_python():
    # Utf8Spec.hs:45-48 "lowerCodePoint is equivalent to Char.toLower on all of Unicode",
    # with python's Unicode 13 simple mapping + the 40 pairs Unicode 14 added (tests/golden/unicode14_lower_additions.json)
    # standing in for Char.toLower.
    for cp in range(0x110000):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        assert oracle.lower_code_point(cp) == ord(naive._lower_cp(chr(cp))), hex(cp)


def test_benchmark_example_file(golden):
    row = golden["benchmark_example_file"]
    m = oracle.Machine(row["needles"])
    assert m.count_matches(0, row["haystack"]) == row["count_naive"]
    assert naive.count_matches(row["needles"], row["haystack"]) == row["count_naive"]


def test_duplicate_needles_value_order():
    # Automaton.hs:263 insertWith (++): duplicates are reported newest-first;
    # Automaton.hs:375: own values, then the fallback state's (shorter suffixes).
    needles = ["b", "ab", "b", "xab"]
    m = oracle.Machine(needles)
    pos, val = m.run_list(0, "xab")
    assert [(int(p), int(v)) for p, v in zip(pos, val)] == [(3, 3), (3, 1), (3, 2), (3, 0)]
    assert naive.all_matches(needles, "xab") == [(3, 3), (3, 1), (3, 2), (3, 0)]


ALPHABETS = ["abAB12", "яЯåÅÅ𝄞💩ßẞ", "aİkKKσΣς"]


def _fragment_case(rng):
    """tests/Data/Text/TestInstances.hs:59-93 arbitraryNeedlesHaystack, restated."""
    alphabet = rng.choice(ALPHABETS)
    frags = ["".join(rng.choice(alphabet) for _ in range(rng.randint(1, 5))) for _ in range(rng.randint(1, 8))]
    needles = ["".join(rng.choice(frags) for _ in range(rng.randint(1, 3))) for _ in range(rng.randint(1, 12))]
    haystack = "".join(rng.choice(frags) for _ in range(rng.randint(1, 60)))
    return needles, haystack


@pytest.mark.parametrize("seed", range(40))
def test_oracle_vs_naive_fragment_pool(seed):
    rng = random.Random(seed)
    for _ in range(25):
        needles, haystack = _fragment_case(rng)
        for case in (0, 1):
            ns = [oracle.lower_utf8(n).decode("utf-8") for n in needles] if case else needles
            m = oracle.Machine(ns)
            # arbitraryOffset (TestInstances.hs:26-33): positions are relative to the slice
            pad = "zz" * rng.randint(0, 3)
            blob = (pad + haystack).encode("utf-8")
            off = len(pad.encode("utf-8"))
            pos, val = m.run_list(case, blob, off, len(blob) - off)
            got = [(int(p), int(v)) for p, v in zip(pos, val)]
            assert got == naive.all_matches(ns, haystack, bool(case)), (needles, haystack, case)


@pytest.mark.parametrize("seed", range(10))
def test_replacer_equals_sequential_replace(seed):
    # AhoCorasickSpec.hs:154-163 (case-sensitive; needles <= 3 chars of "abAB", haystack biased to it)
    rng = random.Random(1000 + seed)
    for _ in range(50):
        pairs = [("".join(rng.choice("abAB") for _ in range(rng.randint(1, 3))),
                  "".join(rng.choice("abABxyİ") for _ in range(rng.randint(0, 4)))) for _ in range(rng.randint(0, 5))]
        haystack = "".join(rng.choice("abAB" * 10 + "İz") for _ in range(rng.randint(0, 40)))
        r = oracle.Replacer(0, pairs)
        assert r.run(haystack).decode("utf-8") == naive.sequential_replace(pairs, haystack), (pairs, haystack)


@pytest.mark.parametrize("seed", range(5))
def test_replacer_compose_property(seed):
    # AhoCorasickSpec.hs:137-148: run (compose a b) == run b . run a ; compose = needles1 ++ needles2
    # with renumbered priorities (Replacer.hs:120-133).
    rng = random.Random(2000 + seed)
    for _ in range(40):
        mk = lambda: [("".join(rng.choice("abAB") for _ in range(rng.randint(1, 3))),
                       "".join(rng.choice("abAB") for _ in range(rng.randint(0, 3)))) for _ in range(rng.randint(0, 4))]
        p1, p2 = mk(), mk()
        haystack = "".join(rng.choice("abAB" * 10 + "İ") for _ in range(rng.randint(0, 30)))
        for case in (0, 1):
            r1, r2, r12 = oracle.Replacer(case, p1), oracle.Replacer(case, p2), oracle.Replacer(case, p1 + p2)
            assert r2.run(r1.run(haystack)) == r12.run(haystack), (p1, p2, haystack, case)


def test_replacer_limit():
    r = oracle.Replacer(0, [("a", "bbbb")])
    assert r.run("aa", max_len=8) == b"bbbbbbbb"
    assert r.run("aa", max_len=7) is None


def test_synthetic_vectors_match_oracle():
    """The committed synthetic vectors (tests/golden/synthetic_vectors.json) are reproduced by the oracle."""
    import json, os
    from tests.conftest import ROOT
    from tests.helpers import oracle_triples
    data = json.load(open(os.path.join(ROOT, "tests", "golden", "synthetic_vectors.json")))
    assert len(data["cases"]) >= 100
    for c in data["cases"]:
        m = oracle.Machine(c["needles"])
        assert oracle_triples(m, c["case"], c["haystacks"]) == [tuple(t) for t in c["triples"]]
