"""Product host layer (C++ mirror of Automaton.build, alfred-margaret_amd/host/automaton.hpp) vs the
oracle: the packed arrays must be bit-identical to the reference layout.  CPU only: build() flattens
on the host; nothing here launches a kernel."""
import random

import numpy as np
import pytest

import alfred_margaret_amd as am
from oracle import oracle
from tests.helpers import fragment_case


def _same_machine(needles):
    p, o = am.Automaton(needles), oracle.Machine(needles)
    assert p.n_states == o.n_states
    assert np.array_equal(p.transitions(), o.transitions())
    assert np.array_equal(p.offsets(), o.offsets())
    assert np.array_equal(p.root_ascii(), o.root_ascii())
    assert np.array_equal(p.values_off(), o.values_off())
    assert np.array_equal(p.values(), o.values())


def test_build_golden_needle_sets(golden):
    for row in golden["count_matches"] + golden["contains_any"] + golden["match_lists"]:
        if row["needles"]:
            _same_machine(row["needles"])
    # 3 needles -> 17 states, 33 packed entries (16 gotos + 17 wildcards)
    a = am.Automaton(["tshirt", "shirts", "shorts"])
    assert a.n_states == 17 and len(a.transitions()) == 33


@pytest.mark.parametrize("seed", range(20))
def test_build_fragment_pool(seed):
    rng = random.Random(seed)
    for _ in range(30):
        needles, _ = fragment_case(rng)
        _same_machine(needles)


def test_build_duplicates_and_empty():
    _same_machine(["b", "ab", "b", "xab", "", "b", ""])
    _same_machine([""])
    _same_machine([])


def test_build_large_random():
    rng = random.Random(99)
    alphabet = "abcdefghijklmnopqrstuvwxyz0123456789 éßя"
    needles = ["".join(rng.choice(alphabet) for _ in range(rng.randint(1, 16))) for _ in range(20000)]
    _same_machine(needles)


def test_lower_and_unlower(golden):
    for row in golden["unlower"]:
        assert sorted(am.unlower_code_point(ord(row["cp"]))) == sorted(ord(c) for c in row["set"]), row["src"]
    for cp in list(range(0x3000)) + [0x10400, 0x1E900, 0x1E921, 0x1F574, 0x10FFFF]:
        assert am.lower_code_point(cp) == oracle.lower_code_point(cp)
    assert am.lower_utf8("GROẞ İK") == oracle.lower_utf8("GROẞ İK")


def test_skip_code_points_backwards(golden):
    for row in golden["skip_code_points_backwards"]:
        if row["expected"] == "error":
            with pytest.raises(IndexError):
                am.skip_code_points_backwards(row["text"], row["index"], row["n"])
        else:
            assert am.skip_code_points_backwards(row["text"], row["index"], row["n"]) == row["expected"], row


def test_create_rejects_malformed_arrays():
    import ctypes as C
    lib = am.api.libam()
    m = oracle.Machine(["ab", "b"])
    tr, of, ra = m.transitions(), m.offsets(), m.root_ascii()
    vl = np.diff(m.values_off()).astype(np.uint32)
    bad = tr.copy()
    bad[0] = (np.uint64(999) << np.uint64(32)) | np.uint64(ord("a"))     # goto to a state that does not exist
    h = C.c_void_p()
    rc = lib.am_automaton_create(bad.ctypes.data, len(bad), of.ctypes.data, m.n_states, ra.ctypes.data, vl.ctypes.data, C.byref(h))
    assert rc == am.AM_ERR_INVALID and b"out of range" in lib.am_last_error()
    rc = lib.am_automaton_create(tr.ctypes.data, len(tr), of.ctypes.data, m.n_states, ra.ctypes.data, vl.ctypes.data, C.byref(h))
    assert rc == am.AM_OK
    lib.am_automaton_destroy(h)
