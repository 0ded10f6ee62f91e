"""Synthetic workloads of BASELINE.json / SURVEY 8d (benchmark and test INPUT generation only).

Needles: deterministic (random.Random(seed)), 4-16 code points, 95 % from [a-z0-9 ] (upper-cased
30 % of the time in mixed-case workloads) and 5 % Latin-1 / Greek / Cyrillic letters, 0.1 % deliberate
duplicates.  IgnoreCase workloads lower-case the needles before build, as Replacer.build does
(reference src/Data/Text/AhoCorasick/Replacer.hs:105-107) -- Searcher.build would not.
Haystacks: csrc/am_synth.h, counter-based per 1-KiB cell, identical on device and host.
"""
import ctypes as C
import os
import random

import numpy as np

from . import api
from . import build as _build

CELL = 1024
NEEDLE_SEED = 0xA1F2ED01
HAYSTACK_SEED = 0xA1F2ED02

WORKLOADS = {
    # name: (n_needles, case, haystack_bytes, n_haystacks, mixed_case)
    "cfg2_runText_10k_1GiB": dict(n_needles=10_000, case=api.CASE_SENSITIVE, hay_bytes=64 << 10, n_hay=16384, mixed=False),
    "cfg3_runLower_100k_10GiB": dict(n_needles=100_000, case=api.IGNORE_CASE, hay_bytes=1 << 20, n_hay=10240, mixed=True),
    "cfg4_100k_1M_haystacks": dict(n_needles=100_000, case=api.IGNORE_CASE, hay_bytes=100 << 10, n_hay=1 << 20, mixed=True, sharded_total=True),
    "cfg5_replacer_50k_1GiB": dict(n_needles=50_000, case=api.CASE_SENSITIVE, hay_bytes=64 << 10, n_hay=16384, mixed=False),
}

_ALPHA = "abcdefghijklmnopqrstuvwxyz0123456789 "
_EXTRA = [chr(c) for c in list(range(0xC0, 0xD7)) + list(range(0xD8, 0xF7)) + list(range(0xF8, 0x100))
          + list(range(0x391, 0x3A2)) + list(range(0x3A3, 0x3AA)) + list(range(0x3B1, 0x3CA))
          + list(range(0x410, 0x450))]


def make_needles(n, mixed_case, seed=NEEDLE_SEED):
    """Returns list[str] of n needles (before any lower-casing)."""
    rng = random.Random(seed)
    out, seen = [], set()
    while len(out) < n:
        if out and rng.random() < 0.001:
            out.append(out[rng.randrange(len(out))])      # deliberate duplicate: exercises value order
            continue
        ln = rng.randint(4, 16)
        chars = []
        for _ in range(ln):
            if rng.random() < 0.05:
                chars.append(rng.choice(_EXTRA))
            else:
                c = rng.choice(_ALPHA)
                if mixed_case and c.isalpha() and rng.random() < 0.3:
                    c = c.upper()
                chars.append(c)
        s = "".join(chars)
        if s in seen:
            continue
        seen.add(s)
        out.append(s)
    return out


def needles_for(workload):
    w = WORKLOADS[workload]
    ns = make_needles(w["n_needles"], w["mixed"])
    if w["case"] == api.IGNORE_CASE:
        ns = [api.lower_utf8(s).decode("utf-8") for s in ns]
    return ns


def replacer_pairs(workload):
    """(needle, replacement) pairs of a Replacer workload (BASELINE config 5): the workload's needles, each with a
    replacement of 0-16 upper-case letters -- an alphabet disjoint from the needles', so passes terminate quickly."""
    w = WORKLOADS[workload]
    needles = make_needles(w["n_needles"], w["mixed"], seed=NEEDLE_SEED + 5)
    rng = np.random.default_rng(5)
    repls = ["".join(chr(ord("A") + int(x)) for x in rng.integers(0, 26, size=int(rng.integers(0, 17)))) for _ in needles]
    return list(zip(needles, repls))


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_build.LIB, "libam_synth.so")
        if not os.path.exists(path):
            path = _build.build_synth()
        _lib = C.CDLL(path)
        _lib.amsynth_generate_device.restype = C.c_int
        _lib.amsynth_generate_device.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                                 C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        _lib.amsynth_generate_host.restype = None
        _lib.amsynth_generate_host.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint32,
                                               C.c_uint64, C.c_uint64, C.c_void_p]
    return _lib


def haystacks_host(needles, mixed_case, first_cell, n_cells, seed=HAYSTACK_SEED):
    """numpy uint8[n_cells * 1024]: cells first_cell .. first_cell + n_cells of the batch."""
    blob, offs = api.pack_texts(needles)
    out = np.zeros(n_cells * CELL, dtype=np.uint8)
    lib().amsynth_generate_host(seed, 1 if mixed_case else 0, CELL, blob, offs.ctypes.data, len(needles), first_cell, n_cells, out.ctypes.data)
    return out


def haystacks_device(needles, mixed_case, first_cell, n_cells, device, seed=HAYSTACK_SEED, pad=64):
    """torch uint8 tensor [n_cells * 1024 (+pad)] generated in HBM; returns (tensor, n_bytes)."""
    import torch
    blob, offs = api.pack_texts(needles)
    d_blob = torch.frombuffer(bytearray(blob + b"\0"), dtype=torch.uint8).to(device)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(device)
    n_bytes = n_cells * CELL
    out = torch.zeros(n_bytes + pad, dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream
    rc = lib().amsynth_generate_device(seed, 1 if mixed_case else 0, CELL, d_blob.data_ptr(), d_offs.data_ptr(), len(needles),
                                       first_cell, n_cells, out.data_ptr(), stream)
    if rc != 0:
        raise RuntimeError("amsynth_generate_device: hip error %d" % rc)
    torch.cuda.synchronize(device)
    return out, n_bytes
