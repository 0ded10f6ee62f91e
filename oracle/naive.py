"""Second, independent oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Does not restate the reference's automaton at all: for every needle it enumerates every
(overlapping) occurrence by brute force over code points, then orders the matches the way the
reference's fold sees them (README.md:87-100 of the reference, Automaton.hs:263,375):
  * ascending end position (code unit index one past the match, relative to the haystack),
  * at one end position longer needles first (own values, then the fallback chain),
  * equal needles in REVERSE insertion order (IntMap.insertWith (++), Automaton.hs:263).
It guards against am_oracle.c restating a misreading.  Same idea as the reference's
benchmark/naive.py (count occurrences per needle) and the `all isInfixOf` properties
(tests/Data/Text/AhoCorasickSpec.hs:202-218).  Empty needles are not supported here (their
quirk, Automaton.hs:502-503/519 + :373-376, is pinned separately by the golden tests).

Lower-casing uses Python's own str.lower() per code point (U+0130 -> U+0069 special-cased:
simple mapping), i.e. it is also independent of the generated C table; the 40 case pairs Unicode 14.0
added (this Python knows 13.0) come from the data file tests/golden/unicode14_lower_additions.json.
"""
import json
import os

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "unicode14_lower_additions.json")) as _f:
    _ADDED = {chr(a): chr(b) for a, b in json.load(_f)["pairs"]}


def _lower_cp(ch):
    if ch == "İ":
        return "i"
    if ch in _ADDED:
        return _ADDED[ch]
    low = ch.lower()
    return low if len(low) == 1 else ch


def all_matches(needles, haystack, ignore_case=False):
    """needles: list[str], haystack: str -> list[(end_pos_bytes, needle_index)] in fold order."""
    hay_cps = list(haystack)
    ends = []
    pos = 0
    for ch in hay_cps:
        pos += len(ch.encode("utf-8"))
        ends.append(pos)
    if ignore_case:
        hay_cps = [_lower_cp(c) for c in hay_cps]
    out = []
    for idx, needle in enumerate(needles):
        ncps = list(needle)
        n = len(ncps)
        if n == 0:
            raise ValueError("naive oracle does not model the empty-needle quirk")
        for start in range(0, len(hay_cps) - n + 1):
            if hay_cps[start:start + n] == ncps:
                out.append((ends[start + n - 1], -n, -idx, idx))
    out.sort()
    return [(e, idx) for (e, _, _, idx) in out]


def count_matches(needles, haystack, ignore_case=False):
    return len(all_matches(needles, haystack, ignore_case))


def sequential_replace(pairs, haystack):
    """tests/Data/Text/AhoCorasickSpec.hs:154-163: foldl' Text.replace (case-sensitive)."""
    for needle, repl in pairs:
        haystack = haystack.replace(needle, repl)
    return haystack
