"""Malformed inputs at the C ABI are refused with AM_ERR_INVALID: they neither hang nor reach a kernel (round-1 advisor
findings).  CPU only: the checks run before any device is touched."""
import ctypes as C
import struct

import numpy as np
import pytest

import alfred_margaret_amd as am
from tests.helpers import ImgCheck

WILDCARD = 0x200000


def test_transition_lists_that_share_entries_are_refused():
    """30 states, a root with 29 ASCII edges, every child's offset pointing at the root's wildcard entry (n_transitions = 30):
    every index is in range and every list is wildcard-terminated, but the lists do not partition the array.  Used to make
    am_automaton_create loop forever while filling the goto table."""
    S = 30
    tr = np.zeros(S, dtype=np.uint64)
    for i in range(29):
        tr[i] = (np.uint64(i + 1) << np.uint64(32)) | np.uint64(ord("a") + i if i < 26 else ord("0") + i - 26)
    tr[29] = np.uint64(WILDCARD)                                     # the root's wildcard: fallback 0
    offsets = np.full(S + 1, 29, dtype=np.uint32)
    offsets[0] = 0
    offsets[S] = 30
    root = np.full(128, WILDCARD, dtype=np.uint64)
    for i in range(29):
        root[int(tr[i]) & 0x1FFFFF] = tr[i]
    vlen = np.zeros(S, dtype=np.uint32)
    h = C.c_void_p()
    rc = am.api.libam().am_automaton_create(tr.ctypes.data, len(tr), offsets.ctypes.data, S, root.ctypes.data, vlen.ctypes.data, C.byref(h))
    assert rc == am.AM_ERR_INVALID
    assert b"partition" in am.api.libam().am_last_error()


def test_stale_image_is_refused_even_with_a_valid_checksum():
    """A serialised image whose header parses and whose checksum matches, but whose body points outside its tables (a node's
    child id here), must not be attached."""
    chk = ImgCheck()
    a = am.Automaton(["tshirt", "shirts", "shorts", "a-needle-long-enough-for-the-trie"])
    img = chk.flatten(a, 0).copy()
    hdr = struct.Struct("<4IQ4I" + "7Q" + "2I" + "4I" + "Q" + "4Q" + "4I" + "8Q")
    assert hdr.size <= 312
    f = hdr.unpack_from(img.tobytes())
    # locate sections by name through the checker's documented layout: off_nodes follows tier_log2_cap[4]
    names = ["magic", "version", "case_mode", "flags", "total_bytes", "n_states", "max_needle_cps", "root_vlen", "ac_chunk",
             "off_transitions", "n_transitions", "off_offsets", "off_root_ascii", "off_canon", "off_vlen", "off_lower",
             "n_lower", "sf_enabled", "sf_tiers", "sf_bloom_log2_words", "sf_n_nodes", "ac_goto_log2_cap",
             "off_bloom", "off_tier0", "off_tier1", "off_tier2", "off_tier3", "cap0", "cap1", "cap2", "cap3",
             "off_nodes", "off_edges", "n_edges", "off_edge_maps", "n_edge_maps", "off_t4_slots", "checksum", "off_goto"]
    h = dict(zip(names, f))
    assert h["magic"] == 0x31474D41 and h["sf_n_nodes"] > 4
    lib = am.api.libam()
    good = bytes(img)
    out = C.c_void_p()
    rc = lib.am_automaton_from_host_image(good, len(good), C.byref(out))
    assert rc in (am.AM_OK, am.AM_ERR_NO_DEVICE)                      # intact image: accepted (or no GPU in this container)
    if rc == am.AM_OK:
        lib.am_automaton_destroy(out)
    # corrupt: node 1's child id (SfNode.z at byte 8 of the 32-byte record) -> far outside the node table
    bad = bytearray(good)
    node = h["off_nodes"] + 32 * 1
    w = struct.unpack_from("<I", bad, node + 12)[0]
    struct.pack_into("<I", bad, node + 12, (w & ~0xFFFF) | 1)       # exactly one edge ...
    struct.pack_into("<I", bad, node + 8, 0x7FFFFFFF)                # ... to a node that does not exist
    # recompute the checksum the way the library does (FNV-1a over 8-byte words, then the tail bytes), header excluded
    body = bytes(bad[312:]) if False else None
    hsize = _header_size(lib, good)
    x = 0xcbf29ce484222325
    data = bytes(bad[hsize:])
    n8 = len(data) // 8
    for wv in struct.unpack_from("<%dQ" % n8, data):
        x = ((x ^ wv) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    for bv in data[n8 * 8:]:
        x = ((x ^ bv) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    struct.pack_into("<Q", bad, _checksum_offset(good, h["checksum"]), x)
    rc = lib.am_automaton_from_host_image(bytes(bad), len(bad), C.byref(out))
    assert rc == am.AM_ERR_INVALID, (rc, lib.am_last_error())
    assert b"image:" in lib.am_last_error()


def _checksum_offset(good, value):
    i = good.find(struct.pack("<Q", value))
    assert 0 < i < 400
    return i


def _header_size(lib, good):
    # the body starts right after the header; the first section is 256-byte aligned, so the header size is what the checksum
    # routine skips: sizeof(ImageHeader).  Recover it by finding which prefix length reproduces the stored checksum.
    for size in range(200, 400, 8):
        x = 0xcbf29ce484222325
        data = good[size:]
        n8 = len(data) // 8
        for wv in struct.unpack_from("<%dQ" % n8, data):
            x = ((x ^ wv) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        for bv in data[n8 * 8:]:
            x = ((x ^ bv) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        if good.find(struct.pack("<Q", x)) in range(0, size):
            return size
    raise AssertionError("header size not found")


def _restamp(lib, good, bad):
    """`bad` (a modified copy of the image `good`) with the checksum the library would compute for it."""
    hsize = _header_size(lib, good)
    stored = struct.unpack_from("<Q", good, _checksum_offset(good, _stored_checksum(good, hsize)))[0]
    x = 0xcbf29ce484222325
    data = bytes(bad[hsize:])
    n8 = len(data) // 8
    for wv in struct.unpack_from("<%dQ" % n8, data):
        x = ((x ^ wv) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    for bv in data[n8 * 8:]:
        x = ((x ^ bv) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    struct.pack_into("<Q", bad, _checksum_offset(good, stored), x)
    return bytes(bad)


def _stored_checksum(good, hsize):
    x = 0xcbf29ce484222325
    data = good[hsize:]
    n8 = len(data) // 8
    for wv in struct.unpack_from("<%dQ" % n8, data):
        x = ((x ^ wv) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    for bv in data[n8 * 8:]:
        x = ((x ^ bv) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return x


def test_damaged_dfa_section_is_refused():
    """Image version 13: the table-walk kernel follows `next`, `out`, `cls`, `fail` and the rare-edge hash of the DFA section without bounds checks, so an
    image from outside whose section points out of its tables (or whose fallbacks never reach the root) must be refused like a damaged suffix section."""
    chk = ImgCheck()
    chk.set("AM_DFA", 1)
    chk.set("AM_DFA_RARE_PERMILLE", 300)
    try:
        good = bytes(chk.flatten(am.Automaton(["tshirt", "shirts", "shorts", "übergrößen", "shirt"]), 1))
    finally:
        chk.set("AM_DFA", -1)
        chk.set("AM_DFA_RARE_PERMILLE", -1)
    d = chk.dfa_header(np.frombuffer(good, dtype=np.uint8))
    assert d["n_states"] > 20 and d["rare_log2_cap"] >= 4
    lib = am.api.libam()
    out = C.c_void_p()
    rc = lib.am_automaton_from_host_image(good, len(good), C.byref(out))
    assert rc in (am.AM_OK, am.AM_ERR_NO_DEVICE), lib.am_last_error()
    if rc == am.AM_OK:
        lib.am_automaton_destroy(out)

    def refused(mutate, what):
        bad = bytearray(good)
        mutate(bad)
        rc = lib.am_automaton_from_host_image(_restamp(lib, good, bad), len(bad), C.byref(out))
        assert rc == am.AM_ERR_INVALID, (what, rc, lib.am_last_error())
        assert b"DFA" in lib.am_last_error(), (what, lib.am_last_error())

    refused(lambda b: struct.pack_into("<I", b, d["off_next"] + 4 * 5, d["n_states"] + 7), "a transition to a state that does not exist")
    refused(lambda b: struct.pack_into("<I", b, d["off_next"] + 4 * 5, struct.unpack_from("<I", b, d["off_next"] + 4 * 5)[0] ^ 0x80000000), "a needle-end bit that disagrees with the target")
    refused(lambda b: struct.pack_into("<B", b, d["off_cls"] + ord("s"), 200), "a byte class without a column")
    refused(lambda b: struct.pack_into("<I", b, d["off_out"] + 8 * 3, 10 ** 6), "a needle end at a reference state that does not exist")
    refused(lambda b: (struct.pack_into("<I", b, d["off_fail"] + 4 * 7, 9), struct.pack_into("<I", b, d["off_fail"] + 4 * 9, 7)), "fallbacks that go round in a circle")
    refused(lambda b: struct.pack_into("<I", b, d["off_fail"], 3), "a root that falls back")
    assert d["n_rows"] < d["n_states"], "the test automaton has chain states"
    refused(lambda b: struct.pack_into("<I", b, d["off_chain"] + 4, (struct.unpack_from("<I", b, d["off_chain"] + 4)[0] & 0xFF000000) | (d["n_rows"] + 1)), "a chain state that falls back to a state without a row")
    refused(lambda b: struct.pack_into("<I", b, d["off_chain"], d["n_states"] + 3), "a chain state whose child does not exist")
    # image version 17: a record leans on a ROW state of its own chain of fallbacks; two-entry records
    n_two = d["n_states"] - d["n_rows"] - d["n_single"]
    assert d["n_single"] > 0 and n_two > 0, "the test automaton has records of both kinds"
    lean = struct.unpack_from("<I", good, d["off_chain"] + 4)[0]
    fail = struct.unpack_from("<%dI" % d["n_states"], good, d["off_fail"])
    on_chain = set()
    s = d["n_rows"]
    while s:
        s = fail[s]; on_chain.add(s)
    stranger = next(r for r in range(d["n_rows"]) if r not in on_chain)
    refused(lambda b: struct.pack_into("<I", b, d["off_chain"] + 4, (lean & 0xFF000000) | stranger), "a record that leans on a row state it never falls back to")
    refused(lambda b: struct.pack_into("<I", b, d["off_chain2"] + 8, d["n_states"] + 1), "a two-entry record whose second target does not exist")
    refused(lambda b: struct.pack_into("<I", b, d["off_chain2"] + 12, struct.unpack_from("<I", b, d["off_chain2"] + 4)[0] & 0xFF000000), "a two-entry record with the same class twice")
    refused(lambda b: struct.pack_into("<I", b, d["off_chain2"] + 4, struct.unpack_from("<I", b, d["off_chain2"] + 4)[0] | 0x00FFFFFF), "a two-entry record that leans on a state without a row")
