"""ctypes front-end over libam.so (C ABI, include/am.h) and libam_host.so (C++ host mirror).

The classes mirror the reference modules (same names and argument meaning):
  Automaton  ~ Data.Text.AhoCorasick.Automaton  (build, run_with_case / run_text / run_lower, count_matches)
  Searcher   ~ Data.Text.AhoCorasick.Searcher   (build, contains_any, contains_all, set_case_sensitivity)
  Replacer   ~ Data.Text.AhoCorasick.Replacer   (build, run, run_with_limit)
Every match position comes from the HIP kernels; there is no Python or CPU matching path, and
importing this module fails loudly if the native libraries are missing and cannot be built.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

CASE_SENSITIVE = 0
IGNORE_CASE = 1

AM_OK = 0
AM_ERR_INVALID, AM_ERR_NO_DEVICE, AM_ERR_HIP, AM_ERR_OOM, AM_ERR_UNSUPPORTED = -1, -2, -3, -4, -5


class AmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libam error %d: %s" % (code, msg))
        self.code = code


class Slice(C.Structure):          # am_slice
    _fields_ = [("ptr", C.c_void_p), ("off", C.c_size_t), ("len", C.c_size_t)]


MATCH_DTYPE = np.dtype([("end_pos", np.uint64), ("haystack", np.uint32), ("state", np.uint32)])   # am_match
PRIO_MATCH_DTYPE = np.dtype([("start", np.uint64), ("len", np.uint64), ("haystack", np.uint32), ("payload", np.uint32)])   # am_prio_match

_u8p, _u32p, _u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_vp, _sz = C.c_void_p, C.c_size_t

# name -> (restype, argtypes): every symbol include/am.h declares
ABI = {
    "am_last_error": (C.c_char_p, []),
    "am_automaton_create": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _vp, C.POINTER(_vp)]),
    "am_automaton_create_ex": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp, _sz, C.POINTER(_vp)]),
    "am_automaton_lower_hash": (C.c_uint32, [_vp]),
    "am_lower_table_hash": (C.c_uint32, [_vp, _vp, _sz]),
    "am_automaton_destroy": (None, [_vp]),
    "am_automaton_set_kernel": (C.c_int, [_vp, C.c_int]),
    "am_count": (C.c_int, [_vp, C.c_int, C.POINTER(Slice), _sz, _vp]),
    "am_contains_any": (C.c_int, [_vp, C.c_int, C.POINTER(Slice), _sz, _vp]),
    "am_run": (C.c_int, [_vp, C.c_int, C.POINTER(Slice), _sz, C.POINTER(_vp)]),
    "am_run_range": (C.c_int, [_vp, C.c_int, C.POINTER(Slice), C.c_uint64, C.c_uint64, C.POINTER(_vp)]),
    "am_count_range": (C.c_int, [_vp, C.c_int, C.POINTER(Slice), C.c_uint64, C.c_uint64, _u64p]),
    "am_batch_upload": (C.c_int, [C.POINTER(Slice), _sz, C.POINTER(_vp)]),
    "am_batch_from_device": (C.c_int, [_vp, _vp, _sz, C.c_uint64, C.POINTER(_vp)]),
    "am_batch_destroy": (None, [_vp]),
    "am_batch_total_bytes": (C.c_uint64, [_vp]),
    "am_count_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _u64p]),
    "am_contains_any_batch": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "am_run_batch": (C.c_int, [_vp, C.c_int, _vp, C.POINTER(_vp)]),
    "am_matches_size": (C.c_uint64, [_vp]),
    "am_matches_data": (_vp, [_vp]),
    "am_matches_device_data": (_vp, [_vp]),
    "am_matches_haystack_range": (C.c_int, [_vp, C.c_uint32, _u64p, _u64p]),
    "am_matches_copy": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp]),
    "am_matches_free": (None, [_vp]),
    "am_matches_fold_hash": (C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "am_needle_ids_create": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(_vp)]),
    "am_needle_ids_destroy": (None, [_vp]),
    "am_contains_all": (C.c_int, [_vp, C.c_int, C.POINTER(Slice), _sz, _vp]),
    "am_contains_all_batch": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "am_replacer_create": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _sz, _vp, _sz, C.c_int64, C.POINTER(_vp)]),
    "am_replacer_destroy": (None, [_vp]),
    "am_replacer_run": (C.c_int, [_vp, C.POINTER(Slice), _sz, C.c_uint64, C.POINTER(_vp)]),
    "am_replacer_run_batch": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(_vp)]),
    "am_replacer_run_batch_device": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(_vp)]),
    "am_replaced_device": (C.c_int, [_vp]),
    "am_replaced_read": (C.c_int, [_vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "am_run_priority": (C.c_int, [_vp, C.POINTER(Slice), _sz, _vp, _vp, C.POINTER(_vp), C.POINTER(_sz)]),
    "am_prio_matches_free": (None, [_vp]),
    "am_replaced_size": (C.c_uint64, [_vp]),
    "am_replaced_get": (C.c_int, [_vp, _sz, C.POINTER(_vp), C.POINTER(_sz)]),
    "am_replaced_passes": (C.c_uint64, [_vp]),
    "am_replaced_scanned_bytes": (C.c_uint64, [_vp]),
    "am_replaced_spliced_bytes": (C.c_uint64, [_vp]),
    "am_replaced_free": (None, [_vp]),
    "am_automaton_image_size": (C.c_int, [_vp, C.c_int, C.POINTER(_sz)]),
    "am_automaton_image_copy": (C.c_int, [_vp, C.c_int, _vp, _sz]),
    "am_automaton_from_image": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "am_automaton_image_read": (C.c_int, [_vp, C.c_int, _vp, _sz]),
    "am_automaton_from_host_image": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "am_multi_unique_id": (C.c_int, [_vp]),
    "am_multi_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "am_multi_create_rank": (C.c_int, [C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "am_multi_destroy": (None, [_vp]),
    "am_multi_local_devices": (C.c_int, [_vp]),
    "am_multi_world_size": (C.c_int, [_vp]),
    "am_multi_device": (C.c_int, [_vp, C.c_int]),
    "am_multi_broadcast_automaton": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "am_multi_allreduce_sum": (C.c_int, [_vp, _vp, _sz]),
    "am_multi_count": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.POINTER(Slice), _sz, _vp, _u64p]),
    "am_multi_run": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.POINTER(Slice), _sz, C.POINTER(_vp), C.POINTER(_sz)]),
    "am_multi_matches_free": (None, [_vp]),
    "am_multi_batch_upload": (C.c_int, [_vp, C.c_int, C.POINTER(Slice), _sz, C.POINTER(_vp)]),
    "am_multi_count_batch": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.POINTER(_vp), C.POINTER(_vp), _u64p, _u64p]),
    "am_multi_run_batch": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.POINTER(_vp), C.POINTER(_vp), _u64p]),
    "am_multi_count_single": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.POINTER(Slice), _u64p, _u64p]),
    "am_multi_run_single": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.POINTER(Slice), C.POINTER(_vp), C.POINTER(_sz), _u64p]),
    "am_lower_code_point": (C.c_uint32, [C.c_uint32]),
    "am_unicode_version": (C.c_uint32, []),
    "am_image_version": (C.c_uint32, []),
    "am_unlower_code_point": (_sz, [C.c_uint32, _vp, _sz]),
    "am_set_stream": (C.c_int, [_vp]),
    "am_get_stream": (C.c_int, [C.POINTER(_vp)]),
    "am_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(_sz), C.c_char_p, _sz]),
    "am_release_host_memory": (C.c_int, []),
    "am_release_device_memory": (C.c_int, []),
    "am_profile_enable": (C.c_int, [C.c_int]),
    "am_profile_reset": (C.c_int, []),
    "am_profile_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), _u64p]),
}

_HOST = {
    "amh_last_error": (C.c_char_p, []),
    "amh_build": (C.c_int, [C.c_char_p, _vp, _sz, _vp, C.POINTER(_vp)]),
    "amh_build_ex": (C.c_int, [C.c_char_p, _vp, _sz, _vp, _vp, _vp, _sz, C.POINTER(_vp)]),
    "amh_free": (None, [_vp]),
    "amh_num_states": (_sz, [_vp]),
    "amh_num_transitions": (_sz, [_vp]),
    "amh_transitions": (_u64p, [_vp]),
    "amh_offsets": (_u32p, [_vp]),
    "amh_root_ascii": (_u64p, [_vp]),
    "amh_values_off": (_u64p, [_vp]),
    "amh_values": (_u32p, [_vp]),
    "amh_device": (_vp, [_vp]),
    "amh_run_list": (C.c_int, [_vp, C.c_int, C.POINTER(Slice), _sz, _vp, _vp, _vp, C.c_uint64, _u64p]),
    "amh_count": (C.c_int, [_vp, C.c_int, C.POINTER(Slice), _sz, _vp]),
    "amh_searcher_build": (C.c_int, [C.c_int, C.c_char_p, _vp, _sz, C.POINTER(_vp)]),
    "amh_searcher_free": (None, [_vp]),
    "amh_searcher_set_case": (None, [_vp, C.c_int]),
    "amh_searcher_contains_any": (C.c_int, [_vp, C.POINTER(Slice), _sz, _vp]),
    "amh_searcher_contains_all": (C.c_int, [_vp, C.POINTER(Slice), _sz, _vp]),
    "amh_searcher_contains_all_host_fold": (C.c_int, [_vp, C.POINTER(Slice), _sz, _vp]),
    "amh_replacer_build": (C.c_int, [C.c_int, C.c_char_p, _vp, C.c_char_p, _vp, _sz, C.POINTER(_vp)]),
    "amh_replacer_build_ex": (C.c_int, [C.c_int, C.c_char_p, _vp, C.c_char_p, _vp, _sz, _vp, _vp, _sz, C.POINTER(_vp)]),
    "amh_replacer_free": (None, [_vp]),
    "amh_replacer_with_replacements": (C.c_int, [_vp, C.c_char_p, _vp, C.POINTER(_vp)]),
    "amh_replacer_set_case": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "amh_replacer_compose": (C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    "amh_replacer_run_batch": (C.c_int, [_vp, C.POINTER(Slice), _sz, C.c_longlong, C.POINTER(_vp), _vp, _vp]),
    "amh_replacer_run_batch_host_splice": (C.c_int, [_vp, C.POINTER(Slice), _sz, C.c_longlong, C.POINTER(_vp), _vp, _vp]),
    "amh_replacer_last_stats": (None, [_vp, _vp, _vp]),
    "amh_replacer_device": (C.c_int, [_vp, C.POINTER(_vp)]),
    "amh_free_blob": (None, [_vp]),
    "amh_splitter_build": (C.c_int, [C.c_char_p, _sz, C.POINTER(_vp)]),
    "amh_splitter_free": (None, [_vp]),
    "amh_splitter_split_batch": (C.c_int, [_vp, C.c_int, C.POINTER(Slice), _sz, C.POINTER(_vp), C.POINTER(_vp), _u64p, _vp]),
    "amh_free_u64": (None, [_vp]),
    "amh_skip_code_points_backwards": (C.c_int64, [C.c_char_p, _sz, _sz, _sz]),
    "amh_lower_utf8": (_sz, [C.c_char_p, _sz, _vp, _sz]),
    "amh_is_case_invariant": (C.c_int, [C.c_char_p, _sz, _vp, _vp, _sz]),
    "amh_needle_casings": (C.c_int, [C.c_char_p, _sz, _vp, _vp, _sz, C.POINTER(_vp), C.POINTER(_vp), _u64p]),
}

# include/am_debug.h: tests and measurements only
DEBUG_ABI = {
    "am_debug_set": (C.c_int, [C.c_char_p, C.c_long]),
    "am_debug_pinned_bytes": (C.c_uint64, []),
    "am_debug_bounds_report": (C.c_int, [_u64p, _u32p, _u32p]),
    "am_debug_sf_phase_cycles": (C.c_int, [_vp]),
    "am_debug_sf_wave_records": (C.c_int, [_vp, _sz]),
    "am_debug_resident_waves": (C.c_int, [_vp, _vp]),
    "am_debug_set_general_kernel": (C.c_int, [_vp, C.c_uint32]),
    "am_debug_rp_lds_haystacks": (C.c_uint32, []),
}

_libam = None
_libhost = None
_libcheck = None


def _bind(lib, table):
    for name, (rt, at) in table.items():
        fn = getattr(lib, name)      # AttributeError = missing export: fail loudly
        fn.restype = rt
        fn.argtypes = at
    return lib


def libam():
    """The product library.  Built in-tree by build.py (hipcc --offload-arch=gfx950); never replaced by a fallback."""
    global _libam
    if _libam is None:
        path = os.path.join(_build.LIB, "libam.so")
        if not os.path.exists(path):
            path = _build.build_libam()
        _libam = _bind(_bind(C.CDLL(path, mode=C.RTLD_GLOBAL), ABI), DEBUG_ABI)
    return _libam


def load_check():
    """TEST INFRASTRUCTURE: loads libam_check.so (tests/native/am_ac.hip), which hands k_ac -- the general AC-walk kernel, the parity gate's
    independent second algorithm -- to libam; Automaton.set_kernel(1) works afterwards.  Called by tests/conftest.py, bench.py's parity gate
    and __graft_entry__.smoke(); nothing in this package calls it."""
    global _libcheck
    if _libcheck is None:
        libam()
        path = os.path.join(_build.LIB, "libam_check.so")
        if not os.path.exists(path):
            path = _build.build_check()
        lib = C.CDLL(path)
        lib.am_check_registered.restype = C.c_int
        if lib.am_check_registered() != 1:
            raise RuntimeError("libam_check.so could not register k_ac with libam (built against another image version?)")
        _libcheck = lib
    return _libcheck


def libhost():
    global _libhost
    if _libhost is None:
        libam()
        path = os.path.join(_build.LIB, "libam_host.so")
        if not os.path.exists(path):
            path = _build.build_host()
        _libhost = _bind(C.CDLL(path), _HOST)
    return _libhost


def check(rc):
    if rc != AM_OK:
        raise AmError(rc, (libam().am_last_error() or b"").decode("utf-8", "replace"))


def _hcheck(rc):
    if rc != AM_OK:
        raise AmError(rc, (libhost().amh_last_error() or b"").decode("utf-8", "replace"))


def _as_bytes(t):
    return t.encode("utf-8") if isinstance(t, str) else bytes(t)


def pack_texts(texts):
    bs = [_as_bytes(t) for t in texts]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum(np.fromiter(map(len, bs), dtype=np.uint64, count=len(bs)))
    return b"".join(bs), offs


class _Slices:
    """Keeps the Python buffers alive while C borrows them.  Accepts str/bytes or (bytes, off, len)."""

    def __init__(self, texts):
        self.keep = []
        self.n = len(texts)
        self.arr = (Slice * max(self.n, 1))()
        for i, t in enumerate(texts):
            off, ln = 0, None
            if isinstance(t, tuple):
                t, off, ln = t
            if isinstance(t, np.ndarray):
                assert t.dtype == np.uint8 and t.flags["C_CONTIGUOUS"]
                ptr, size = t.ctypes.data, t.size
                self.keep.append(t)
            else:
                b = _as_bytes(t)
                buf = C.create_string_buffer(b, len(b) + 1)
                self.keep.append(buf)
                ptr, size = C.addressof(buf), len(b)
            self.arr[i] = Slice(ptr, off, size - off if ln is None else ln)


class _LowerArrays:
    """The caller's lower-case table as the two arrays the C ABI takes.  None = the built-in table (NULL, NULL, 0); an EMPTY list is a
    table of its own -- ASCII-only lower-casing -- and travels as two valid pointers with n = 0, as am_automaton_create_ex reads it."""

    def __init__(self, pairs):
        self.n = 0 if pairs is None else len(pairs)
        self.f = self.t = None
        if pairs is not None:
            self.f = np.zeros(max(self.n, 1), dtype=np.uint32)
            self.t = np.zeros(max(self.n, 1), dtype=np.uint32)
            for i, (a, b) in enumerate(pairs):
                self.f[i], self.t[i] = a, b
        self.from_ptr = None if self.f is None else self.f.ctypes.data
        self.to_ptr = None if self.t is None else self.t.ctypes.data


def lower_table_hash(pairs=None):
    la = _LowerArrays(pairs)
    return int(libam().am_lower_table_hash(la.from_ptr, la.to_ptr, la.n))


class Automaton:
    """AcMachine v with v = uint32 handles (needle index by default)."""

    def __init__(self, needles, values=None, lower_pairs=None):
        """lower_pairs: the caller's lower-casing as [(c, toLower c)] (am_automaton_create_ex); None = the built-in Unicode 14.0 table."""
        self.needles = [_as_bytes(n) for n in needles]            # (encoded once: 100k needles are 20 ms of the build per pass over them)
        blob, offs = pack_texts(self.needles)
        vptr = None
        if values is not None:
            self._vals = np.ascontiguousarray(values, dtype=np.uint32)
            vptr = self._vals.ctypes.data
        h = _vp()
        la = _LowerArrays(lower_pairs)
        _hcheck(libhost().amh_build_ex(blob, offs.ctypes.data, len(needles), vptr, la.from_ptr, la.to_ptr, la.n, C.byref(h)))
        self._h = h

    @property
    def lower_hash(self):
        return int(libam().am_automaton_lower_hash(self.device))

    @classmethod
    def build(cls, needles_with_values):
        """Automaton.hs:176 build :: [(Text, v)] -> AcMachine v (v = uint32)."""
        return cls([n for n, _ in needles_with_values], [v for _, v in needles_with_values])

    def __del__(self):
        if getattr(self, "_h", None):
            libhost().amh_free(self._h)
            self._h = None

    @property
    def device(self):
        return libhost().amh_device(self._h)

    @property
    def n_states(self):
        return libhost().amh_num_states(self._h)

    def transitions(self):
        return np.ctypeslib.as_array(libhost().amh_transitions(self._h), shape=(libhost().amh_num_transitions(self._h),)).copy()

    def offsets(self):
        return np.ctypeslib.as_array(libhost().amh_offsets(self._h), shape=(self.n_states + 1,)).copy()

    def root_ascii(self):
        return np.ctypeslib.as_array(libhost().amh_root_ascii(self._h), shape=(128,)).copy()

    def values_off(self):
        return np.ctypeslib.as_array(libhost().amh_values_off(self._h), shape=(self.n_states + 1,)).copy()

    def values(self):
        n = int(self.values_off()[-1])
        return np.ctypeslib.as_array(libhost().amh_values(self._h), shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    def set_kernel(self, k):
        check(libam().am_automaton_set_kernel(self.device, k))

    def image_bytes(self, case):
        """The flattened automaton of one case mode as a serialisable blob (am_automaton_image_read)."""
        n = C.c_size_t(0)
        check(libam().am_automaton_image_size(self.device, case, C.byref(n)))
        buf = (C.c_uint8 * n.value)()
        check(libam().am_automaton_image_read(self.device, case, buf, n.value))
        return bytes(buf)

    def save_image(self, path, case):
        with open(path, "wb") as f:
            f.write(self.image_bytes(case))

    def run_batch_with_case(self, case, texts):
        """Per-haystack list fold of runWithCase: returns (haystack, matchPos, value) arrays in fold order."""
        s = _Slices(texts)
        n = C.c_uint64(0)
        _hcheck(libhost().amh_run_list(self._h, case, s.arr, s.n, None, None, None, 0, C.byref(n)))
        k = int(n.value)
        hay, pos, val = np.zeros(max(k, 1), np.uint32), np.zeros(max(k, 1), np.uint64), np.zeros(max(k, 1), np.uint32)
        _hcheck(libhost().amh_run_list(self._h, case, s.arr, s.n, hay.ctypes.data, pos.ctypes.data, val.ctypes.data, k, C.byref(n)))
        return hay[:k], pos[:k], val[:k]

    def run_with_case(self, case, text):
        _, pos, val = self.run_batch_with_case(case, [text])
        return pos, val

    def run_text(self, text):
        return self.run_with_case(CASE_SENSITIVE, text)

    def run_lower(self, text):
        return self.run_with_case(IGNORE_CASE, text)

    def count_matches(self, case, texts):
        """countMatches (benchmark/haskell/app/Main.hs:67-76) per haystack."""
        s = _Slices(texts)
        out = np.zeros(max(s.n, 1), np.uint64)
        _hcheck(libhost().amh_count(self._h, case, s.arr, s.n, out.ctypes.data))
        return out[:s.n]

    def run_records(self, case, texts):
        """Raw am_match records straight from the C ABI (am_run)."""
        s = _Slices(texts)
        m = _vp()
        check(libam().am_run(self.device, case, s.arr, s.n, C.byref(m)))
        try:
            return matches_to_numpy(m)
        finally:
            libam().am_matches_free(m)


class ValuesTable:
    """machineValues of an Automaton in flat form on the device (am_needle_ids): what the fold-checksum and
    containsAll kernels expand records with."""

    def __init__(self, automaton, n_needles=None):
        self._a = automaton                      # the am_automaton must outlive the table
        self._off = np.ascontiguousarray(automaton.values_off(), dtype=np.uint64)
        self._val = np.ascontiguousarray(automaton.values(), dtype=np.uint32)
        if self._val.size == 0:
            self._val = np.zeros(1, np.uint32)
        h = _vp()
        n = len(automaton.needles) if n_needles is None else n_needles
        check(libam().am_needle_ids_create(automaton.device, self._off.ctypes.data, self._val.ctypes.data, n, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            libam().am_needle_ids_destroy(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def fold_hash(self, matches, n_hay):
        """am_matches_fold_hash: (hash[n_hay], count[n_hay]) of the fold sequences of an am_matches* result."""
        hashes, counts = np.zeros(max(n_hay, 1), np.uint64), np.zeros(max(n_hay, 1), np.uint64)
        check(libam().am_matches_fold_hash(matches, self._h, n_hay, hashes.ctypes.data, counts.ctypes.data))
        return hashes[:n_hay], counts[:n_hay]


class ImageAutomaton:
    """A device automaton attached to a serialised image (am_automaton_from_host_image): no build, no
    flatten.  It serves the image's case mode and returns raw records; machineValues stay with whoever
    built the automaton."""

    def __init__(self, image_bytes):
        buf = bytes(image_bytes)
        h = _vp()
        check(libam().am_automaton_from_host_image(buf, len(buf), C.byref(h)))
        self._h = h

    @classmethod
    def load(cls, path):
        with open(path, "rb") as f:
            return cls(f.read())

    def __del__(self):
        if getattr(self, "_h", None):
            libam().am_automaton_destroy(self._h)
            self._h = None

    @property
    def device(self):
        return self._h.value

    def run_records(self, case, texts):
        s = _Slices(texts)
        m = _vp()
        check(libam().am_run(self._h, case, s.arr, s.n, C.byref(m)))
        try:
            return matches_to_numpy(m)
        finally:
            libam().am_matches_free(m)

    def count_matches(self, case, texts):
        s = _Slices(texts)
        out = np.zeros(max(s.n, 1), np.uint64)
        check(libam().am_count(self._h, case, s.arr, s.n, out.ctypes.data))
        return out[:s.n]


def matches_to_numpy(m):
    n = int(libam().am_matches_size(m))
    if n == 0:
        return np.zeros(0, MATCH_DTYPE)
    p = libam().am_matches_data(m)
    if not p:
        raise AmError(AM_ERR_HIP, (libam().am_last_error() or b"").decode())
    return np.frombuffer((C.c_char * (n * MATCH_DTYPE.itemsize)).from_address(p), dtype=MATCH_DTYPE).copy()


def matches_of_haystack(m, haystack):
    """The records of ONE haystack of a (possibly huge, device-resident) result as a numpy array: binary search + one small copy (am_matches_haystack_range / am_matches_copy)."""
    first, count = C.c_uint64(0), C.c_uint64(0)
    check(libam().am_matches_haystack_range(m, int(haystack), C.byref(first), C.byref(count)))
    out = np.zeros(count.value, MATCH_DTYPE)
    if count.value:
        check(libam().am_matches_copy(m, first.value, count.value, out.ctypes.data))
    return out


class Searcher:
    def __init__(self, case, needles):
        blob, offs = pack_texts(needles)
        h = _vp()
        _hcheck(libhost().amh_searcher_build(case, blob, offs.ctypes.data, len(needles), C.byref(h)))
        self._h = h
        self.case = case

    build = classmethod(lambda cls, case, needles: cls(case, needles))

    def __del__(self):
        if getattr(self, "_h", None):
            libhost().amh_searcher_free(self._h)
            self._h = None

    def set_case_sensitivity(self, case):
        libhost().amh_searcher_set_case(self._h, case)
        self.case = case

    def contains_any_batch(self, texts):
        s = _Slices(texts)
        out = np.zeros(max(s.n, 1), np.uint8)
        _hcheck(libhost().amh_searcher_contains_any(self._h, s.arr, s.n, out.ctypes.data))
        return out[:s.n].astype(bool)

    def contains_any(self, text):
        return bool(self.contains_any_batch([text])[0])

    def contains_all_batch(self, texts, host_fold=False):
        """Searcher.containsAll per haystack.  host_fold=True folds the records on the host (cross-check)."""
        s = _Slices(texts)
        out = np.zeros(max(s.n, 1), np.uint8)
        fn = libhost().amh_searcher_contains_all_host_fold if host_fold else libhost().amh_searcher_contains_all
        _hcheck(fn(self._h, s.arr, s.n, out.ctypes.data))
        return out[:s.n].astype(bool)

    def contains_all(self, text):
        return bool(self.contains_all_batch([text])[0])


class Replacer:
    def __init__(self, case, pairs, lower_pairs=None):
        nb, no = pack_texts([p[0] for p in pairs])
        rb, ro = pack_texts([p[1] for p in pairs])
        h = _vp()
        la = _LowerArrays(lower_pairs)
        _hcheck(libhost().amh_replacer_build_ex(case, nb, no.ctypes.data, rb, ro.ctypes.data, len(pairs), la.from_ptr, la.to_ptr, la.n, C.byref(h)))
        self._h = h
        self.pairs = list(pairs)

    build = classmethod(lambda cls, case, pairs: cls(case, pairs))

    @classmethod
    def _wrap(cls, handle, pairs):
        r = cls.__new__(cls)
        r._h = handle
        r.pairs = pairs
        return r

    def map_replacement(self, f):
        """Replacer.mapReplacement (Replacer.hs:135-141): new replacements, same needles, no rebuild."""
        fresh = [f(_as_bytes(rep)) for _, rep in self.pairs]
        rb, ro = pack_texts(fresh)
        h = _vp()
        _hcheck(libhost().amh_replacer_with_replacements(self._h, rb, ro.ctypes.data, C.byref(h)))
        return Replacer._wrap(h, [(n, rep) for (n, _), rep in zip(self.pairs, fresh)])

    @staticmethod
    def compose(first, second):
        """Replacer.compose (Replacer.hs:120-133): `second` after `first` as ONE replacer (the second's priorities renumbered to
        come after the first's); None when the case sensitivities differ."""
        h = _vp()
        _hcheck(libhost().amh_replacer_compose(first._h, second._h, C.byref(h)))
        return Replacer._wrap(h, first.pairs + second.pairs) if h.value else None

    def set_case_sensitivity(self, case):
        """Replacer.setCaseSensitivity (Replacer.hs:148-153): needles untouched."""
        h = _vp()
        _hcheck(libhost().amh_replacer_set_case(self._h, case, C.byref(h)))
        return Replacer._wrap(h, self.pairs)

    def __del__(self):
        if getattr(self, "_h", None):
            libhost().amh_replacer_free(self._h)
            self._h = None

    def run_batch(self, texts, max_len=-1, host_splice=False):
        """Replacer.run / runWithLimit on a batch.  host_splice=True keeps only the scans on the GPU
        (fold, sort and splice on the host) -- an independent cross-check of the device passes."""
        s = _Slices(texts)
        blob = _vp()
        offs = np.zeros(s.n + 1, np.uint64)
        nothing = np.zeros(max(s.n, 1), np.uint8)
        fn = libhost().amh_replacer_run_batch_host_splice if host_splice else libhost().amh_replacer_run_batch
        _hcheck(fn(self._h, s.arr, s.n, max_len, C.byref(blob), offs.ctypes.data, nothing.ctypes.data))
        try:
            raw = C.string_at(blob, int(offs[-1]))
        finally:
            libhost().amh_free_blob(blob)
        return [None if nothing[i] else raw[int(offs[i]):int(offs[i + 1])] for i in range(s.n)]

    @property
    def device(self):
        """am_replacer* for the raw ABI (am_replacer_run_batch on device-resident batches)."""
        h = _vp()
        _hcheck(libhost().amh_replacer_device(self._h, C.byref(h)))
        return h.value

    def run_priority(self, texts, thresholds):
        """One pass of the Replacer fold on the device (am_run_priority): (best priority per haystack, selected matches)."""
        s = _Slices(texts)
        thr = np.ascontiguousarray(thresholds, dtype=np.int64)
        best = np.zeros(max(s.n, 1), np.int64)
        p, n = _vp(), _sz(0)
        check(libam().am_run_priority(self.device, s.arr, s.n, thr.ctypes.data, best.ctypes.data, C.byref(p), C.byref(n)))
        try:
            k = int(n.value)
            ms = np.frombuffer((C.c_char * (k * PRIO_MATCH_DTYPE.itemsize)).from_address(p.value), dtype=PRIO_MATCH_DTYPE).copy() if k else np.zeros(0, PRIO_MATCH_DTYPE)
        finally:
            libam().am_prio_matches_free(p)
        return best[:s.n], ms

    def last_stats(self):
        """(passes, haystack bytes scanned over all passes) of the last run_batch."""
        a = np.zeros(2, np.uint64)
        libhost().amh_replacer_last_stats(self._h, a[0:].ctypes.data, a[1:].ctypes.data)
        return int(a[0]), int(a[1])

    def run(self, text):
        return self.run_batch([text])[0]

    def run_with_limit(self, max_len, text):
        return self.run_batch([text], max_len)[0]


class Splitter:
    """Data.Text.AhoCorasick.Splitter (build :64-67, split :84-85, splitIgnoreCase :96-97)."""

    def __init__(self, separator):
        b = _as_bytes(separator)
        h = _vp()
        _hcheck(libhost().amh_splitter_build(b, len(b), C.byref(h)))
        self._h = h

    build = classmethod(lambda cls, separator: cls(separator))

    def __del__(self):
        if getattr(self, "_h", None):
            libhost().amh_splitter_free(self._h)
            self._h = None

    def split_batch(self, texts, ignore_case=False):
        s = _Slices(texts)
        blob, offs = _vp(), _vp()
        nf = C.c_uint64(0)
        per = np.zeros(max(s.n, 1), np.uint32)
        _hcheck(libhost().amh_splitter_split_batch(self._h, 1 if ignore_case else 0, s.arr, s.n, C.byref(blob), C.byref(offs), C.byref(nf), per.ctypes.data))
        try:
            o = np.ctypeslib.as_array(C.cast(offs, _u64p), shape=(nf.value + 1,)).copy()
            raw = C.string_at(blob, int(o[-1]))
        finally:
            libhost().amh_free_blob(blob)
            libhost().amh_free_u64(offs)
        out, k = [], 0
        for i in range(s.n):
            out.append([raw[int(o[k + j]):int(o[k + j + 1])] for j in range(int(per[i]))])
            k += int(per[i])
        return out

    def split(self, text):
        return self.split_batch([text], False)[0]

    def split_ignore_case(self, text):
        return self.split_batch([text], True)[0]


DEBUG_SWITCHES = ("AM_SF_ABLATE", "AM_SF_POOL_BLOCKS", "AM_SF_WQ", "AM_SF_WQ_ITERS", "AM_SF_MAX_BLOOM_LOG2_WORDS", "AM_SF_NO_CHILDREN", "AM_SF_PROBE_TWO", "AM_NO_SMALL_RUN", "AM_DFA", "AM_DFA_CHUNK", "AM_DFA_RARE_PERMILLE", "AM_DFA_MIN_KIB", "AM_DFA_TUNE", "AM_DFA_HOT_LOG2", "AM_DFA_NO_CHAINS", "AM_FLATTEN_TRACE", "AM_FLATTEN_SERIAL", "AM_NO_IDS_SCAN",
                  "AM_RP_FULL_SCANS", "AM_RP_SPLICE", "AM_RP_PIECES", "AM_RP_PARALLEL_FOLD", "AM_RP_GROUPS", "AM_RP_NO_FUSE", "AM_RP_NO_SPIN",
                  "AM_RP_MAT_MAIN", "AM_RP_NO_RANGE_REUSE", "AM_RP_TRACE", "AM_RP_LOOP_WAVES", "AM_RP_LDS", "AM_RP_NO_PLI", "AM_RP_LOOP", "AM_RUN_SEGMENTS")


def debug_set(name, value):
    """A test / measurement switch of libam (csrc/am_config.h) by the name of its environment variable; -1 = unset.  No switch changes a result."""
    check(libam().am_debug_set(name.encode(), int(value)))


def debug_reset():
    for name in DEBUG_SWITCHES:
        debug_set(name, -1)


def resident_waves():
    """am_debug_resident_waves: (wavefronts a CU of the current device really runs at once -- 32 by the architecture, 16 on boxes where a second set of 16 waits for the
    first --, ms of the 16-per-CU launch, ms of the 32-per-CU launch)."""
    one, two = C.c_float(0), C.c_float(0)
    check(libam().am_debug_resident_waves(C.byref(one), C.byref(two)))
    return (32 if two.value < 1.5 * one.value else 16), round(one.value, 3), round(two.value, 3)


def bounds_report():
    """(failed index assertions, line of the first, translation units that carry assertions) of a -DAM_BOUNDS_CHECK build; (0, 0, 0) in the product build."""
    f, l, u = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
    check(libam().am_debug_bounds_report(C.byref(f), C.byref(l), C.byref(u)))
    return int(f.value), int(l.value), int(u.value)


def image_version():
    """kImageVersion of the flattened automaton image this build produces (am_image_version()), so that tools and bench.py can
    refuse measurements taken with another layout."""
    return int(libam().am_image_version())


def lower_code_point(cp):
    return int(libam().am_lower_code_point(cp))


def unlower_code_point(cp):
    buf = (C.c_uint32 * 16)()
    n = libam().am_unlower_code_point(cp, buf, 16)
    return [int(buf[i]) for i in range(n)]


def lower_utf8(text):
    b = _as_bytes(text)
    out = C.create_string_buffer(len(b) * 4 + 4)
    n = libhost().amh_lower_utf8(b, len(b), out, len(b) * 4 + 4)
    return out.raw[:n]


def is_case_invariant(text, lower_pairs=None):
    """Utf8.isCaseInvariant (Utf8.hs:169-171)."""
    b = _as_bytes(text)
    la = _LowerArrays(lower_pairs)
    r = libhost().amh_is_case_invariant(b, len(b), la.from_ptr, la.to_ptr, la.n)
    if r < 0:
        raise AmError(AM_ERR_INVALID, (libhost().amh_last_error() or b"").decode("utf-8", "replace"))
    return bool(r)


def needle_casings(text, lower_pairs=None):
    """Automaton.needleCasings (Automaton.hs:555-566): list[bytes], in the reference's order."""
    b = _as_bytes(text)
    la = _LowerArrays(lower_pairs)
    blob, offs, n = _vp(), _vp(), C.c_uint64(0)
    _hcheck(libhost().amh_needle_casings(b, len(b), la.from_ptr, la.to_ptr, la.n, C.byref(blob), C.byref(offs), C.byref(n)))
    try:
        o = np.ctypeslib.as_array(C.cast(offs, _u64p), shape=(n.value + 1,)).copy()
        raw = C.string_at(blob, int(o[-1]))
    finally:
        libhost().amh_free_blob(blob)
        libhost().amh_free_u64(offs)
    return [raw[int(o[i]):int(o[i + 1])] for i in range(int(n.value))]


def skip_code_points_backwards(text, index, n):
    b = _as_bytes(text)
    r = libhost().amh_skip_code_points_backwards(b, len(b), index, n)
    if r < 0:
        raise IndexError("Invalid use of skipCodePointsBackwards")
    return int(r)


def device_info():
    n_cu, hbm = C.c_int(0), C.c_size_t(0)
    name = C.create_string_buffer(64)
    check(libam().am_device_info(C.byref(n_cu), C.byref(hbm), name, 64))
    return {"n_cu": n_cu.value, "hbm_bytes": hbm.value, "arch": name.value.decode()}
