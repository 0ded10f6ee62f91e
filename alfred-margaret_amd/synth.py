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
    # SURVEY 8d shape (i) of configs[1]: the same gibibyte as ONE haystack (the reference scans one Text of any size, Automaton.hs:468-480)
    "cfg2_single_1GiB": dict(n_needles=10_000, case=api.CASE_SENSITIVE, hay_bytes=1 << 30, n_hay=1, mixed=False),
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


# ---- "natural text" (robustness sweep): a Zipf-distributed vocabulary with shared suffixes; needles are vocabulary words and
# two-word phrases.  Nothing of the reference's (unpublished) data set is reproduced; this only puts the scanner into the
# regime README.md:14-25 describes -- real words, shared stems and suffixes, dense matches.
_ONSETS = ["", "b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r", "s", "t", "v", "w", "br", "ch", "cl", "cr", "dr", "fl", "fr", "gr",
           "pl", "pr", "sh", "sl", "sp", "st", "str", "th", "tr", "wh"]
_VOWELS = ["a", "e", "i", "o", "u", "ai", "ea", "ee", "io", "oo", "ou", "\u00e9", "\u00f6", "\u00fc"]
_CODAS = ["", "", "n", "r", "s", "t", "l", "m", "d", "ng", "nt", "st", "ck", "ll", "ss"]
_SUFFIXES = ["", "", "", "", "s", "ed", "ing", "er", "ers", "tion", "ly", "es", "al", "ment", "able", "ness", "ings", "ation"]
_FOREIGN = ["\u03ba\u03b1\u03b9", "\u03c4\u03bf", "\u03bb\u03cc\u03b3\u03bf\u03c2", "\u0438", "\u043d\u0435", "\u0441\u043b\u043e\u0432\u043e", "\u043c\u0438\u0440", "stra\u00dfe", "\u00e5r"]


class Vocabulary:
    """V words (short ones first = most frequent), p(rank) ~ 1 / (rank + 2.7), and the quantile table the generator samples through."""

    def __init__(self, n_words=1 << 17, n_quantile=1 << 20, seed=0xA1F2ED07):
        rng = random.Random(seed)
        seen, words = set(), []
        for f in _FOREIGN:
            seen.add(f); words.append(f)
        while len(words) < n_words:
            syl = rng.choice((1, 1, 2, 2, 2, 3, 3))
            stem = "".join(rng.choice(_ONSETS) + (rng.choice(_VOWELS[:11]) if rng.random() < 0.97 else rng.choice(_VOWELS[11:])) + rng.choice(_CODAS) for _ in range(syl))
            word = stem + rng.choice(_SUFFIXES)
            if len(word) < 2 or len(word) > 18 or word in seen:
                continue
            seen.add(word); words.append(word)
        keyed = sorted(((len(w) + 3.0 * rng.random(), w) for w in words))
        self.words = [w for _, w in keyed]
        p = 1.0 / (np.arange(n_words, dtype=np.float64) + 2.7)
        cdf = np.cumsum(p / p.sum())
        self.quantile = np.minimum(np.searchsorted(cdf, (np.arange(n_quantile) + 0.5) / n_quantile), n_words - 1).astype(np.uint32)
        self.blob, self.offs = api.pack_texts(self.words)
        self._dev = {}

    def device(self, device):
        import torch
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.frombuffer(bytearray(self.blob + b"\0"), dtype=torch.uint8).to(device),
                              torch.from_numpy(self.offs.astype(np.int64)).to(device),
                              torch.from_numpy(self.quantile.astype(np.int32)).to(device))
        return self._dev[key]

    def needles(self, n, seed=0xA1F2ED08):
        """n distinct needles: 70 % single vocabulary words of at least 4 letters (a fixed pseudo-random 60 % of those ranks, the
        frequent ones included), 30 % two-word phrases whose words are drawn like the text's."""
        rng = random.Random(seed)
        singles = [w for r, w in enumerate(self.words) if len(w) >= 4 and ((r * 2654435761) & 0xFFFFFFFF) < 0x98000000][:int(n * 0.7)]       # keywords, not particles
        seen, out = set(singles), list(singles)
        nq = len(self.quantile)
        while len(out) < n:
            ph = self.words[int(self.quantile[rng.randrange(nq)])] + " " + self.words[int(self.quantile[rng.randrange(nq)])]
            if ph in seen:
                continue
            seen.add(ph); out.append(ph)
        rng.shuffle(out)
        return out


_VOCAB = None


def vocabulary():
    global _VOCAB
    if _VOCAB is None:
        _VOCAB = Vocabulary()
    return _VOCAB


WORKLOADS["natural_100k_10GiB"] = dict(n_needles=100_000, case=api.IGNORE_CASE, hay_bytes=1 << 20, n_hay=10240, mixed=True, natural=True)


def needles_for(workload):
    w = WORKLOADS[workload]
    if w.get("natural"):
        return vocabulary().needles(w["n_needles"])          # lower case already
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
        _lib.amsynth_generate_device_ex.restype = C.c_int
        _lib.amsynth_generate_device_ex.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                                    C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _lib.amsynth_generate_host_ex.restype = None
        _lib.amsynth_generate_host_ex.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p,
                                                  C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _lib.amsynth_generate_host.restype = None
        _lib.amsynth_generate_host.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint32,
                                               C.c_uint64, C.c_uint64, C.c_void_p]
    return _lib


def haystacks_host(needles, mixed_case, first_cell, n_cells, seed=HAYSTACK_SEED, plants=1, natural=False):
    """numpy uint8[n_cells * 1024]: cells first_cell .. first_cell + n_cells of the batch.  plants: needles planted per cell
    (0 = none); natural: Zipf text over vocabulary() instead of random code points."""
    blob, offs = api.pack_texts(needles)
    out = np.zeros(n_cells * CELL, dtype=np.uint8)
    if plants == 1 and not natural:
        lib().amsynth_generate_host(seed, 1 if mixed_case else 0, CELL, blob, offs.ctypes.data, len(needles), first_cell, n_cells, out.ctypes.data)
        return out
    v = vocabulary() if natural else None
    lib().amsynth_generate_host_ex(seed, 1 if mixed_case else 0, CELL, blob, offs.ctypes.data, len(needles) if plants else 0, first_cell, n_cells, out.ctypes.data,
                                   max(plants, 1), 1 if natural else 0, v.blob if v else None, v.offs.ctypes.data if v else None,
                                   v.quantile.ctypes.data if v else None, len(v.quantile) if v else 0)
    return out


def haystacks_device(needles, mixed_case, first_cell, n_cells, device, seed=HAYSTACK_SEED, pad=64, plants=1, natural=False):
    """torch uint8 tensor [n_cells * 1024 (+pad)] generated in HBM; returns (tensor, n_bytes)."""
    import torch
    blob, offs = api.pack_texts(needles)
    d_blob = torch.frombuffer(bytearray(blob + b"\0"), dtype=torch.uint8).to(device)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(device)
    n_bytes = n_cells * CELL
    out = torch.zeros(n_bytes + pad, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        if plants == 1 and not natural:
            rc = lib().amsynth_generate_device(seed, 1 if mixed_case else 0, CELL, d_blob.data_ptr(), d_offs.data_ptr(), len(needles),
                                               first_cell, n_cells, out.data_ptr(), stream)
        else:
            vb, vo, vq = vocabulary().device(device) if natural else (None, None, None)
            rc = lib().amsynth_generate_device_ex(seed, 1 if mixed_case else 0, CELL, d_blob.data_ptr(), d_offs.data_ptr(), len(needles) if plants else 0,
                                                  first_cell, n_cells, out.data_ptr(), stream, max(plants, 1), 1 if natural else 0,
                                                  vb.data_ptr() if natural else None, vo.data_ptr() if natural else None, vq.data_ptr() if natural else None,
                                                  int(vq.numel()) if natural else 0)
        if rc != 0:
            raise RuntimeError("amsynth_generate_device: hip error %d" % rc)
        torch.cuda.synchronize(device)
    return out, n_bytes
