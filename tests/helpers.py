"""Shared helpers for the parity tests (test infrastructure)."""
import ctypes as C
import random
import struct

import numpy as np

from oracle import oracle

ALPHABETS = ["abAB12", "яЯåÅÅ𝄞💩ßẞ", "aİkKKσΣςi", "ab"]


def fragment_case(rng, n_hay_max=5, hay_frags=40, allow_empty_needle=True):
    """tests/Data/Text/TestInstances.hs:46-93 restated: needles and haystacks from a shared fragment pool."""
    alphabet = rng.choice(ALPHABETS)
    frags = ["".join(rng.choice(alphabet) for _ in range(rng.randint(1, 5))) for _ in range(rng.randint(1, 8))]
    needles = ["".join(rng.choice(frags) for _ in range(rng.randint(1, 3))) for _ in range(rng.randint(1, 12))]
    if allow_empty_needle and rng.random() < 0.1:
        needles.append("")
    hays = ["".join(rng.choice(frags) for _ in range(rng.randint(0, hay_frags))) for _ in range(rng.randint(1, n_hay_max))]
    if rng.random() < 0.3:
        hays.insert(rng.randint(0, len(hays)), "")
    return needles, hays


def oracle_triples(machine, case, hays):
    """What the reference's fold sees: [(haystack, matchPos, value)] in fold order."""
    out = []
    for i, h in enumerate(hays):
        pos, val = machine.run_list(case, h)
        out += [(i, int(p), int(v)) for p, v in zip(pos, val)]
    return out


def expand_records(values_off, values, hay, state, end):
    out = []
    for i in range(len(hay)):
        vs = values[int(values_off[state[i]]):int(values_off[state[i] + 1])]
        out += [(int(hay[i]), int(end[i]), int(v)) for v in vs]
    return out


class ImgCheck:
    """ctypes front-end of the TEST-ONLY host interpreter of the device image (libam_imgcheck.so)."""

    def __init__(self):
        from alfred_margaret_amd import build
        self.lib = C.CDLL(build.build_imgcheck())
        self.lib.amchk_flatten.restype = C.c_longlong
        self.lib.amchk_flatten_ex.restype = C.c_longlong
        self.lib.amchk_scan.restype = C.c_longlong

    def flatten(self, m, case, lower_pairs=None):
        """m: anything with transitions()/offsets()/root_ascii()/values_off()/n_states (oracle or product machine).
        lower_pairs: the caller's lower-case table [(c, toLower c)] (am_automaton_create_ex); None = built-in."""
        tr, of, ra = m.transitions(), m.offsets(), m.root_ascii()
        vl = np.diff(m.values_off()).astype(np.uint32)
        err = C.create_string_buffer(256)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        if lower_pairs is None:
            low = (None, None, C.c_size_t(0))
        else:
            lf = np.ascontiguousarray([a for a, _ in lower_pairs], dtype=np.uint32)
            lt = np.ascontiguousarray([b for _, b in lower_pairs], dtype=np.uint32)
            low = (P(lf), P(lt), C.c_size_t(len(lf)))
        args = (P(tr), C.c_size_t(len(tr)), P(of), C.c_size_t(m.n_states), P(ra), P(vl), case) + low
        n = self.lib.amchk_flatten_ex(*args, None, C.c_size_t(0), err, C.c_size_t(256))
        if n < 0:
            raise ValueError(err.value.decode())
        img = np.zeros(n, dtype=np.uint8)
        assert self.lib.amchk_flatten_ex(*args, P(img), C.c_size_t(n), err, C.c_size_t(256)) == n
        return img

    def scan(self, img, which, hays):
        blob, offs = oracle.pack_texts(hays)
        text = np.frombuffer(blob + b"\0", dtype=np.uint8)
        cap = max(16, len(blob) + 16)
        hay, st = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        end, vl = np.zeros(cap, np.uint64), np.zeros(cap, np.uint32)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        n = self.lib.amchk_scan(P(img), which, P(text), P(offs), C.c_uint32(len(hays)), P(hay), P(st), P(end), P(vl), C.c_size_t(cap))
        if n < 0:
            return n, None
        return n, (hay[:n], st[:n], end[:n], vl[:n])

    def set(self, name, value):
        """A switch of csrc/am_config.h inside libam_imgcheck.so's copy of the flattener (AM_DFA, AM_DFA_CHUNK, AM_SF_NO_CHILDREN ...); -1 = unset."""
        self.lib.amchk_set.argtypes = [C.c_char_p, C.c_long]
        assert self.lib.amchk_set(name.encode(), int(value)) == 0, name

    @staticmethod
    def dfa_header(img):
        f = struct.unpack_from("<5Q2IQ4IQ2IQ", img.tobytes()[256:352])      # ImageHeader.off_dfa_next ... off_dfa_chain2
        return {"off_next": f[0], "off_out": f[1], "off_cls": f[2], "off_fail": f[3], "off_rare": f[4], "rare_log2_cap": f[5], "n_rows": f[6], "off_chain": f[7],
                "n_states": f[8], "log2_classes": f[9], "warm": f[10], "chunk": f[11], "off_hot": f[12], "hot_log2": f[13], "n_single": f[14], "off_chain2": f[15]}

    @staticmethod
    def set_dfa_chunk(img, chunk):
        img[324:328] = np.frombuffer(struct.pack("<I", chunk), dtype=np.uint8)   # ImageHeader.dfa_chunk (the host interpreter takes any value >= 1)

    @staticmethod
    def set_ac_chunk(img, chunk):
        img[36:40] = np.frombuffer(struct.pack("<I", chunk), dtype=np.uint8)   # ImageHeader.ac_chunk

    @staticmethod
    def header(img):
        f = struct.unpack_from("<4IQ4I", img.tobytes()[:40])
        return {"magic": f[0], "version": f[1], "case_mode": f[2], "total_bytes": f[4], "n_states": f[5],
                "max_needle_cps": f[6], "root_vlen": f[7], "ac_chunk": f[8]}
