"""Host mirror of the reference's case helpers: Utf8.isCaseInvariant (src/Data/Text/Utf8.hs:169-171), Utf8.unlowerCodePoint's order
(src/Data/Text/Utf8/Unlower.hs:26-40) and Automaton.needleCasings (src/Data/Text/AhoCorasick/Automaton.hs:555-566), against the answers the
reference itself states (tests/Data/Text/Utf8Spec.hs:55-79; the examples in the haddock of needleCasings).  No GPU: lower-casing is table work."""
import itertools

import pytest

import alfred_margaret_amd as am


def test_is_case_invariant_reference_examples():
    # Utf8Spec.hs:64-79
    assert am.is_case_invariant("") is True
    for t in (".", ".,;'123", "\U0001F4A9"):
        assert am.is_case_invariant(t) is True, t
    for t in ("a", "A..", "ß.", "ẞ", "İ"):
        assert am.is_case_invariant(t) is False, t


def test_needle_casings_reference_examples():
    # Automaton.hs:558-560
    assert [c.decode() for c in am.needle_casings("abc")] == ["abc", "abC", "aBc", "aBC", "Abc", "AbC", "ABc", "ABC"]
    assert am.needle_casings("ABC") == []
    assert [c.decode() for c in am.needle_casings("ω1")] == ["Ω1", "ω1", "Ω1"]          # OHM SIGN, omega, Omega
    assert am.needle_casings("") == [b""]
    # unlowerCodePoint 'i' == "İiI", 'ß' == "ẞß", '1' == "1" (Utf8Spec.hs:55-62): highest code point first
    assert [c.decode() for c in am.needle_casings("i")] == ["İ", "i", "I"]
    assert [c.decode() for c in am.needle_casings("ß")] == ["ẞ", "ß"]
    assert am.needle_casings("1") == [b"1"]


def test_every_casing_lowers_back_and_matches_under_ignore_case_semantics():
    for needle in ("kå", "straße", "θx"):
        cs = am.needle_casings(needle)
        assert len(cs) == len(set(cs)) and len(cs) >= 2 ** sum(ch.isalpha() and ch.isascii() for ch in needle)
        for c in cs:
            assert am.lower_utf8(c) == needle.encode(), (needle, c)


def test_caller_table_drives_the_helpers():
    # a caller whose toLower knows one pair only: U+A7DC -> U+019B (Unicode 16; not in the built-in 14.0 table)
    pairs = [(0xA7DC, 0x019B)]
    assert am.is_case_invariant("ƛ") is True
    assert am.is_case_invariant("ƛ", lower_pairs=pairs) is False
    assert [c.decode() for c in am.needle_casings("ƛ", lower_pairs=pairs)] == ["Ƛ", "ƛ"]
    # ... and knows nothing else beyond ASCII: ß has one casing under that table, two under the built-in one
    assert am.needle_casings("ß", lower_pairs=pairs) == ["ß".encode()]
    assert [c.decode() for c in am.needle_casings("aß", lower_pairs=[])] == ["aß", "Aß"]


def test_an_empty_caller_table_is_a_table_of_its_own():
    """lower_pairs=[] is ASCII-only lower-casing (two valid pointers, n = 0 at the C ABI), NOT the built-in table (NULL, NULL, 0)."""
    assert am.lower_table_hash([]) != am.lower_table_hash(None)
    assert am.lower_table_hash([(0x41, 0x61), (0xC4, 0xC4)]) == am.lower_table_hash([])      # ASCII and identity pairs carry no information
    a_builtin, a_empty = am.Automaton(["äb"]), am.Automaton(["äb"], lower_pairs=[])
    assert a_builtin.lower_hash == am.lower_table_hash(None)
    assert a_empty.lower_hash == am.lower_table_hash([])


def test_host_table_refuses_what_the_device_table_refuses():
    with pytest.raises(am.AmError):
        am.Automaton(["x"], lower_pairs=[(0xC4, 0xE4), (0xC4, 0xE5)])          # one code point, two images
    with pytest.raises(am.AmError):
        am.needle_casings("x", lower_pairs=[(0x110000, 0x61)])
    assert am.lower_table_hash([(0xC4, 0xE4), (0xC4, 0xE4)]) == am.lower_table_hash([(0xC4, 0xE4)])
