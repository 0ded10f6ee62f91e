"""Flattener + per-position walk logic (the code the HIP kernels execute, am_image.h) interpreted on
the CPU by the test-only libam_imgcheck.so and compared with the oracle.  CPU only."""
import random

import pytest

import alfred_margaret_amd as am
from oracle import oracle
from tests.helpers import ImgCheck, expand_records, fragment_case, oracle_triples


@pytest.fixture(scope="module")
def chk():
    return ImgCheck()


def _check(chk, needles, hays, case, chunks=(None,)):
    o = oracle.Machine(needles)
    p = am.Automaton(needles)          # product build feeds the flattener, as in production
    img = chk.flatten(p, case)
    exp = oracle_triples(o, case, hays)
    vo, vals = o.values_off(), o.values()
    for which in (0, 1, 2):            # AC walk, SF filter+verify, SF verify-everything
        for chunk in (chunks if which == 0 else (None,)):
            if chunk:
                chk.set_ac_chunk(img, chunk)
            n, recs = chk.scan(img, which, hays)
            if n == -2:
                assert "" in needles   # SF is disabled only for automata with the empty needle
                continue
            assert n >= 0
            assert expand_records(vo, vals, recs[0], recs[1], recs[2]) == exp, (which, chunk, case, needles, hays)
            assert all(int(vo[s + 1] - vo[s]) == int(v) for s, v in zip(recs[1], recs[3]))
    # the table walk (k_dfa's logic) on an image that was made to carry a DFA section (the flattener only gives one to dictionaries by itself)
    if "" not in needles and any(needles):
        chk.set("AM_DFA", 1)
        rare = len("".join(needles)) % 2 == 1                 # every other case: few columns, most bytes take the rare-byte walk (fallback chain + edge hash)
        if rare:
            chk.set("AM_DFA_RARE_PERMILLE", 400)
        try:
            img = chk.flatten(p, case)
        finally:
            chk.set("AM_DFA", -1)
            chk.set("AM_DFA_RARE_PERMILLE", -1)
        assert chk.dfa_header(img)["n_states"] >= 1
        for chunk in chunks:
            if chunk:
                chk.set_dfa_chunk(img, chunk)
            n, recs = chk.scan(img, 3, hays)
            assert n >= 0
            assert expand_records(vo, vals, recs[0], recs[1], recs[2]) == exp, ("dfa", chunk, case, needles, hays)
            assert all(int(vo[s + 1] - vo[s]) == int(v) for s, v in zip(recs[1], recs[3]))


def test_golden_counts_and_lists(chk, golden):
    for row in golden["count_matches"] + golden["match_lists"] + golden["contains_any"]:
        if row["needles"]:
            _check(chk, row["needles"], [row["haystack"]], 0 if row["case"] == "CaseSensitive" else 1)


@pytest.mark.parametrize("seed", range(12))
def test_fragment_pool(chk, seed):
    rng = random.Random(seed)
    for _ in range(40):
        needles, hays = fragment_case(rng)
        for case in (0, 1):
            ns = [oracle.lower_utf8(n).decode() for n in needles] if (case and rng.random() < 0.8) else needles
            _check(chk, ns, hays, case)


@pytest.mark.parametrize("seed", range(6))
def test_chunk_boundaries(chk, seed):
    # tiny AC chunks force unit boundaries inside code points, matches and warm-up regions
    rng = random.Random(100 + seed)
    for _ in range(10):
        needles, hays = fragment_case(rng, n_hay_max=4, hay_frags=300)
        for case in (0, 1):
            ns = [oracle.lower_utf8(n).decode() for n in needles] if case else needles
            _check(chk, ns, hays, case, chunks=(1, 3, 16, 64))


def test_header_fields(chk):
    p = am.Automaton(["abc", "", "xyzw"])
    h = chk.header(chk.flatten(p, 1))
    assert h["magic"] == 0x31474D41 and h["case_mode"] == 1 and h["n_states"] == p.n_states
    assert h["max_needle_cps"] == 4 and h["root_vlen"] == 1 and h["ac_chunk"] >= 256


@pytest.mark.parametrize("seed", range(4))
def test_long_needles_compressed_edges(chk, seed):
    # needles far longer than one compressed edge (16 skip bytes) with shared tails and needles that
    # are suffixes of other needles: exercises label splitting, mid-chain terminals and the
    # haystack-start clipping of the 16-byte label compare
    rng = random.Random(500 + seed)
    for _ in range(8):
        alphabet = rng.choice(["ab", "abcdefgh", "aéß𝄞", "aAkKK"])
        base = ["".join(rng.choice(alphabet) for _ in range(rng.randint(20, 90))) for _ in range(4)]
        needles = []
        for b in base:
            needles.append(b)
            for _ in range(3):
                needles.append(b[rng.randint(1, len(b) - 4):])                    # suffix of a longer needle
            needles.append("".join(rng.choice(alphabet) for _ in range(rng.randint(1, 6))) + b[len(b) // 2:])   # shared tail, other head
        hays = []
        for _ in range(4):
            parts = []
            for _ in range(rng.randint(1, 6)):
                parts.append(rng.choice(needles) if rng.random() < 0.7 else "".join(rng.choice(alphabet) for _ in range(rng.randint(1, 30))))
            hays.append("".join(parts))
        hays.append(needles[0][3:])        # starts inside a needle: clipped at the haystack start
        for case in (0, 1):
            ns = [oracle.lower_utf8(n).decode() for n in needles] if case else needles
            _check(chk, ns, hays, case)


def test_single_haystack_split_across_ranks(chk):
    """dist.split_single_haystack: scanning overlapping ranges and keeping each rank's own end
    positions gives exactly the whole-haystack result, also with cuts inside code points and matches."""
    import numpy as np
    from alfred_margaret_amd import dist as amdist
    rng = random.Random(77)
    for it in range(12):
        needles, hays = fragment_case(rng, n_hay_max=1, hay_frags=400)
        text = max(hays, key=len) if hays else ""
        b = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        for case in (0, 1):
            ns = [oracle.lower_utf8(n).decode() for n in needles] if case else needles
            p = am.Automaton(ns)
            img = chk.flatten(p, case)
            max_cps = max([len(n) for n in ns] + [1])
            for which in (0, 1):        # 0 = general AC walk, 1 = suffix filter (+ the dense part when the empty needle is there)
                _split_equals_whole(chk, img, which, b, max_cps, (1, 2, 3, 7), (case, ns))


def _split_equals_whole(chk, img, which, b, max_cps, worlds, ctx):
    import numpy as np
    from alfred_margaret_amd import dist as amdist
    n, whole = chk.scan(img, which, [b])
    assert n >= 0
    exp = sorted(zip(whole[2].tolist(), whole[1].tolist()))
    for world in worlds:
        got = []
        for start, lo, hi, scan_hi in amdist.split_single_haystack(b, world, max_cps):
            n, part = chk.scan(img, which, [b[start:scan_hi]])
            assert n >= 0
            recs = np.zeros(len(part[0]), dtype=am.api.MATCH_DTYPE)
            recs["end_pos"], recs["state"] = part[2], part[1]
            own = amdist.own_records(recs, start, lo, hi)
            got += list(zip(own["end_pos"].tolist(), own["state"].tolist()))
        assert got == exp, (world, which, ctx, b)


def test_single_haystack_split_inside_code_points_general_kernel(chk):
    """A rank's range may end inside a code point.  The general kernel must not see the truncated sequence (its
    guarded decode would read it as another code point): needle U+00C0 / U+00E0 over U+00E9 x 10 used to report
    matches that do not exist (round-1 advisor finding)."""
    for case, needle in ((0, "\u00c0"), (1, "\u00e0"), (0, "\u00e9"), (1, "\u00e9")):
        p = am.Automaton([needle])
        img = chk.flatten(p, case)
        for hay in ("\u00e9" * 10, "\u00c9\u00e9" * 7, "a\u20ac\U0001d11e\u00e9" * 5):
            for which in (0, 1):
                _split_equals_whole(chk, img, which, hay.encode("utf-8"), 1, (2, 3, 5, 7, 8), (case, needle))


def test_empty_needle_through_the_suffix_filter(chk):
    """With the empty needle among the needles the reference folds the root's values wherever the automaton is not at
    the root after a code point, i.e. wherever some needle PREFIX ends (Automaton.hs:373-376,502-503,519).  The suffix
    filter gets there with extra terminals for prefixes whose last code point starts no needle (the blank of "new york")
    plus a per-position test for first code points; the image must not fall back to the general kernel for these."""
    cases = [
        (["", "new york", "abc"], ["new york", "a new yor", "ab", "xnew  y", "", "abcabc new", "w york"]),
        (["", "a b c d"], ["a b c d", "a b c", " b c d", "aa  b"]),
        (["", "été fini", "k1"], ["ÉTÉ FINI", "été f", "K1 k", "é"]),
        (["", "", "x"], ["xx", "y"]),
        ([""], ["anything", ""]),
    ]
    for needles, hays in cases:
        for case in (0, 1):
            o = oracle.Machine(needles)
            p = am.Automaton(needles)
            img = chk.flatten(p, case)
            exp = oracle_triples(o, case, hays)
            for which in (1, 2, 0):
                n, recs = chk.scan(img, which, hays)
                assert n >= 0, (which, needles)
                assert expand_records(o.values_off(), o.values(), recs[0], recs[1], recs[2]) == exp, (which, case, needles, hays)


@pytest.mark.parametrize("seed", range(6))
def test_wide_fan_out_below_the_suffix(chk, seed):
    """Reversed-needle trie nodes with 5..40 children deeper than the 4-byte suffix (needles that share a stem and differ in the
    letter before it, two levels): the walk finds the child through the node's selector map (SfEdgeMap) and one 64-byte edge line
    that carries the child's record.  Filter + probe + resolve and resolve-everything against the oracle, both case modes."""
    rng = random.Random(300 + seed)
    letters = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
    for _ in range(5):
        stem = "".join(rng.choice("xyz") for _ in range(rng.randint(4, 12)))
        fan = rng.sample(letters, rng.randint(5, 40))
        needles = [c + stem for c in fan]
        for c in fan[:5]:
            needles += [d + c + stem for d in rng.sample(letters, rng.randint(5, 20))]
        needles += [stem, "q" + stem[1:]]
        hays = ["".join(rng.choice(letters[:30]) for _ in range(5)) + n + "".join(rng.choice("xyz") for _ in range(3)) for n in rng.sample(needles, min(len(needles), 40))]
        hays.append(" ".join(needles)[:5000])
        for case in (0, 1):
            ns = list(dict.fromkeys(oracle.lower_utf8(n).decode() for n in needles)) if case else needles
            o, p = oracle.Machine(ns), am.Automaton(ns)
            img = chk.flatten(p, case)
            exp = oracle_triples(o, case, hays)
            vo, vals = o.values_off(), o.values()
            for which in (1, 2):
                n, recs = chk.scan(img, which, hays)
                assert n >= 0
                assert expand_records(vo, vals, recs[0], recs[1], recs[2]) == exp, (case, which, needles[:6])


def test_heavy_suffix_nodes_get_five_byte_child_entries():
    """Round 5 (am_image.h kT4Heavy): in a dictionary of natural-language words the 4-byte suffixes branch as a rule ("tion", "ing ", "ness": dozens of
    different bytes before them; 100k words share 12k suffixes), and a branching depth-4 node used to defer every position with its suffix.  The flattener
    now gives such nodes one hot entry per child under the five-byte key, the probe asks for them where a heavy node is all that speaks for a position,
    and phase 2 starts from the child's slot line.  On the CPU (the image interpreter = the kernels' own probe / resolve code): the records are the
    oracle's through both scan modes, and the probe defers a quarter fewer positions than the suffix alone would."""
    import ctypes as C
    import struct
    from alfred_margaret_amd import synth
    needles = synth.needles_for("natural_100k_10GiB")
    m = oracle.Machine(needles)
    chk = ImgCheck()
    img = chk.flatten(m, 1)
    n_children, = struct.unpack_from("<I", img[:256].tobytes(), 248)          # ImageHeader::sf_t4_children
    assert n_children > 5_000, n_children
    text = bytes(synth.haystacks_host(needles, True, 5, 192, natural=True))
    hays = [text[:65536], text[65536:65536 + 70000], b"", text[140000:]]
    exp = oracle_triples(m, 1, hays)
    chk.lib.amchk_stats.restype = None
    stats = (C.c_uint64 * 3)()
    chk.lib.amchk_stats(stats, 1)
    for which in (1, 2):
        n, recs = chk.scan(img, which, hays)
        hay, st, end, vl = recs
        assert expand_records(m.values_off(), m.values(), hay, st, end) == exp, which
    chk.lib.amchk_stats(stats, 1)
    kib = len(text) / 1024.0
    cand, deferred, found = (stats[i] / kib for i in range(3))
    assert found > 120 and deferred < 0.7 * cand and deferred < 270, (cand, deferred, found)      # 403 candidates, 328 deferred without the child entries, 242 with them
    # the benchmark automata (random needles: branching suffixes are the exception) get none
    img3 = chk.flatten(oracle.Machine(synth.needles_for("cfg3_runLower_100k_10GiB")[:20000]), 1)
    assert struct.unpack_from("<I", img3[:256].tobytes(), 248)[0] == 0


def test_the_flatteners_tasks_change_no_byte(chk):
    """Round 6: the goto hash and the DFA section are made on threads of their own next to the suffix structure, and the walk that weighs a dictionary's DFA states runs in
    stretches on several threads (am_flatten.cpp; the 20 000-word dictionary below is large enough for that).  The image must not know:
    byte for byte the one a flatten without tasks (AM_FLATTEN_SERIAL) makes -- for fragment automata that are forced to carry a DFA section, and for a dictionary large
    enough to get one by itself."""
    import numpy as np
    from alfred_margaret_amd import synth
    rng = random.Random(99)
    cases = []
    for _ in range(25):
        needles, _hays = fragment_case(rng)
        if "" not in needles and any(needles):
            cases.append((needles, True))
    cases.append((synth.vocabulary().needles(20000), False))
    for needles, force in cases:
        a = am.Automaton(needles)
        if force:
            chk.set("AM_DFA", 1)
        try:
            for case in (0, 1):
                chk.set("AM_FLATTEN_SERIAL", 1)
                serial = chk.flatten(a, case)
                chk.set("AM_FLATTEN_SERIAL", -1)
                tasks = chk.flatten(a, case)
                assert np.array_equal(serial, tasks), (len(needles), case)
                if force:
                    assert chk.dfa_header(serial)["n_states"] >= 1
        finally:
            chk.set("AM_DFA", -1)
            chk.set("AM_FLATTEN_SERIAL", -1)
    assert chk.dfa_header(chk.flatten(am.Automaton(cases[-1][0]), 1))["n_states"] > 1000      # (the dictionary got its section unasked: the speculative task was the one kept)
