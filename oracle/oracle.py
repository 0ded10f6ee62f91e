"""ctypes front-end of oracle/am_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The API mirrors the reference modules so that tests read like the reference's own tests:
  Automaton: build / run_list (runText|runLower with a list-building fold) / count_matches
  Searcher : contains_any / contains_all
  Replacer : Replacer(case, pairs).run(text[, max_len])
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("AM_ORACLE_LIB") or os.path.join(_HERE, "_build", "libam_oracle.so")       # (AM_ORACLE_LIB: tools/sanitize.sh's instrumented build)

CASE_SENSITIVE = 0
IGNORE_CASE = 1


def build_library():
    """Compile the C restatement (gcc only; called by __graft_entry__.build())."""
    if not os.environ.get("AM_ORACLE_LIB"):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def _load():
    if not os.path.exists(_LIB_PATH):
        build_library()
    lib = C.CDLL(_LIB_PATH)
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    lib.orc_build.restype = C.c_void_p
    lib.orc_build.argtypes = [C.c_char_p, u64p, C.c_size_t, u32p]
    lib.orc_free.argtypes = [C.c_void_p]
    for name, rt in (("orc_num_states", C.c_size_t), ("orc_num_transitions", C.c_size_t),
                     ("orc_transitions", u64p), ("orc_offsets", u32p), ("orc_root_ascii", u64p),
                     ("orc_values_off", u64p), ("orc_values", u32p), ("orc_fallback", u32p)):
        getattr(lib, name).restype = rt
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.orc_count_matches.restype = C.c_uint64
    lib.orc_count_matches.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t]
    lib.orc_run_list.restype = C.c_uint64
    lib.orc_run_list.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, u64p, u32p, C.c_uint64]
    lib.orc_fold_hash.restype = C.c_uint64
    lib.orc_fold_hash.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, u64p]
    lib.orc_contains_any.restype = C.c_int
    lib.orc_contains_any.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t]
    lib.orc_contains_all.restype = C.c_int
    lib.orc_contains_all.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t]
    lib.orc_set_lower_table.restype = None
    lib.orc_set_lower_table.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.orc_lower_code_point.restype = C.c_uint32
    lib.orc_lower_code_point.argtypes = [C.c_uint32]
    lib.orc_skip_code_points_backwards.restype = C.c_int64
    lib.orc_skip_code_points_backwards.argtypes = [C.c_char_p, C.c_size_t, C.c_int64, C.c_int64]
    lib.orc_lower_utf8.restype = C.c_void_p
    lib.orc_lower_utf8.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.orc_replacer_build.restype = C.c_void_p
    lib.orc_replacer_build.argtypes = [C.c_int, C.c_char_p, u64p, C.c_char_p, u64p, C.c_size_t]
    lib.orc_replacer_free.argtypes = [C.c_void_p]
    lib.orc_replacer_set_case.restype = None
    lib.orc_replacer_set_case.argtypes = [C.c_void_p, C.c_int]
    lib.orc_replacer_run.restype = C.c_void_p
    lib.orc_replacer_run.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64, C.POINTER(C.c_size_t)]
    lib.orc_free_bytes.argtypes = [C.c_void_p]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _as_bytes(t):
    return t.encode("utf-8") if isinstance(t, str) else bytes(t)


def pack_texts(texts):
    """[bytes] -> (concatenated bytes, uint64 offsets[n+1])."""
    bs = [_as_bytes(t) for t in texts]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return b"".join(bs), offs


def _buf_ptr(data):
    """Pointer to the bytes of `data` (bytes or a C-contiguous uint8 numpy array)."""
    if isinstance(data, np.ndarray):
        assert data.dtype == np.uint8 and data.flags["C_CONTIGUOUS"]
        return data.ctypes.data, data.size
    b = _as_bytes(data)
    return C.cast(C.c_char_p(b), C.c_void_p).value, len(b), b


class Machine:
    """Restated `AcMachine v` with v = uint32 (Automaton.hs:108-123)."""

    def __init__(self, needles, values=None):
        blob, offs = pack_texts(needles)
        self.n_needles = len(needles)
        vptr = None
        if values is not None:
            varr = np.ascontiguousarray(values, dtype=np.uint32)
            vptr = varr.ctypes.data_as(C.POINTER(C.c_uint32))
        self._h = lib().orc_build(blob, offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(needles), vptr)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_free(self._h)
            self._h = None

    # packed arrays (copies)
    @property
    def n_states(self):
        return lib().orc_num_states(self._h)

    def transitions(self):
        n = lib().orc_num_transitions(self._h)
        return np.ctypeslib.as_array(lib().orc_transitions(self._h), shape=(n,)).copy()

    def offsets(self):
        return np.ctypeslib.as_array(lib().orc_offsets(self._h), shape=(self.n_states + 1,)).copy()

    def root_ascii(self):
        return np.ctypeslib.as_array(lib().orc_root_ascii(self._h), shape=(128,)).copy()

    def values_off(self):
        return np.ctypeslib.as_array(lib().orc_values_off(self._h), shape=(self.n_states + 1,)).copy()

    def values(self):
        off = self.values_off()
        n = int(off[-1])
        if n == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(lib().orc_values(self._h), shape=(n,)).copy()

    def fallback(self):
        return np.ctypeslib.as_array(lib().orc_fallback(self._h), shape=(self.n_states,)).copy()

    # runs
    def _args(self, text, off, length):
        p = _buf_ptr(text)
        n = p[1]
        if length is None:
            length = n - off
        return p, off, length

    def count_matches(self, case, text, off=0, length=None):
        p, off, length = self._args(text, off, length)
        return int(lib().orc_count_matches(self._h, case, p[0], off, length))

    def run_list(self, case, text, off=0, length=None):
        """Matches in fold order: [(matchPos, value)], positions relative to the slice start."""
        p, off, length = self._args(text, off, length)
        n = int(lib().orc_run_list(self._h, case, p[0], off, length, None, None, 0))
        pos = np.zeros(max(n, 1), dtype=np.uint64)
        val = np.zeros(max(n, 1), dtype=np.uint32)
        lib().orc_run_list(self._h, case, p[0], off, length,
                           pos.ctypes.data_as(C.POINTER(C.c_uint64)), val.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return pos[:n], val[:n]

    def fold_hash(self, case, text, off=0, length=None):
        """(hash, count) of the fold sequence: h' = h * P + mix(matchPos, value), see am_oracle.c orc_fold_hash."""
        p, off, length = self._args(text, off, length)
        n = C.c_uint64(0)
        h = lib().orc_fold_hash(self._h, case, p[0], off, length, C.byref(n))
        return int(h), int(n.value)

    def contains_any(self, case, text, off=0, length=None):
        p, off, length = self._args(text, off, length)
        return bool(lib().orc_contains_any(self._h, case, p[0], off, length))

    def contains_all(self, case, text, off=0, length=None):
        p, off, length = self._args(text, off, length)
        return bool(lib().orc_contains_all(self._h, case, self.n_needles, p[0], off, length))


def set_lower_table(pairs=None):
    """Give the oracle the caller's Data.Char.toLower as [(c, toLower c)] (what am_automaton_create_ex takes); None = back to the
    built-in Unicode 14.0 table.  Process-wide: tests reset it in a finally block."""
    if not pairs:
        lib().orc_set_lower_table(None, None, 0)
        return
    ps = sorted((int(a), int(b)) for a, b in pairs if int(a) >= 128 and int(a) != int(b))
    f = np.ascontiguousarray([a for a, _ in ps], dtype=np.uint32)
    t = np.ascontiguousarray([b for _, b in ps], dtype=np.uint32)
    lib().orc_set_lower_table(f.ctypes.data, t.ctypes.data, len(ps))


def builtin_lower_pairs():
    """The built-in table as data: [(c, lower c)] over all of Unicode (with whatever table is active)."""
    out = []
    for cp in range(128, 0x110000):
        lo = int(lib().orc_lower_code_point(cp))
        if lo != cp:
            out.append((cp, lo))
    return out


def lower_code_point(cp):
    return int(lib().orc_lower_code_point(cp))


def lower_utf8(text):
    b = _as_bytes(text)
    n = C.c_size_t(0)
    p = lib().orc_lower_utf8(b, len(b), C.byref(n))
    out = C.string_at(p, n.value)
    lib().orc_free_bytes(p)
    return out


def skip_code_points_backwards(text, index, n):
    b = _as_bytes(text)
    r = lib().orc_skip_code_points_backwards(b, len(b), index, n)
    if r < 0:
        raise IndexError("Invalid use of skipCodePointsBackwards")
    return int(r)


class Replacer:
    """Restated Data.Text.AhoCorasick.Replacer (build :97-116, run :200-201, runWithLimit :203-274)."""

    def __init__(self, case, pairs):
        nb, no = pack_texts([p[0] for p in pairs])
        rb, ro = pack_texts([p[1] for p in pairs])
        self._h = lib().orc_replacer_build(case, nb, no.ctypes.data_as(C.POINTER(C.c_uint64)),
                                           rb, ro.ctypes.data_as(C.POINTER(C.c_uint64)), len(pairs))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:      # module globals may be gone at interpreter exit
            lib().orc_replacer_free(self._h)
            self._h = None

    def set_case_sensitivity(self, case):
        """Replacer.setCaseSensitivity (Replacer.hs:148-153), in place: needles and payload lengths untouched."""
        lib().orc_replacer_set_case(self._h, int(case))
        return self

    def run(self, text, max_len=-1):
        b = _as_bytes(text)
        n = C.c_size_t(0)
        p = lib().orc_replacer_run(self._h, b, len(b), max_len, C.byref(n))
        if not p:
            return None
        out = C.string_at(p, n.value)
        lib().orc_free_bytes(p)
        return out
