"""libam.so loads, exports every symbol include/am.h declares, and refuses to compute without a GPU."""
import ctypes as C
import os
import re

import pytest

import alfred_margaret_amd as am
from tests.conftest import ROOT


def _declared(header="am.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(am_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 25
    lib = C.CDLL(os.path.join(am.build.LIB, "libam.so"))
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(am.api.ABI) == names      # the Python binding table covers the header exactly
    assert sorted(am.api.DEBUG_ABI) == _declared("am_debug.h")


def test_the_library_exports_the_two_headers_and_nothing_else():
    """libam.so is linked with -fvisibility=hidden + csrc/libam.map: its dynamic symbol table is include/am.h + include/am_debug.h -- no C++
    internals, no kernel handles, and no k_ac (the checker kernel lives in libam_check.so) or oracle symbol."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(am.build.LIB, "libam.so")], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == sorted(_declared() + _declared("am_debug.h")), set(exported) ^ set(_declared() + _declared("am_debug.h"))
    assert not [s for s in exported if "k_ac" in s or "orc_" in s or "launch_" in s]


def test_general_kernel_is_refused_without_the_checker_library():
    """In a process that did not load libam_check.so -- every product process -- am_automaton_set_kernel(a, 1) leads to AM_ERR_UNSUPPORTED, never to
    a silent other path.  (Runs in a child: this process has the checker loaded by conftest.)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import ctypes as C, alfred_margaret_amd as am\n"
            "lib = am.api.libam(); a = am.Automaton(['abc']); a.set_kernel(1)\n"
            "b = C.c_void_p(); s = (am.api.Slice * 1)(); buf = C.create_string_buffer(b'xabcx'); s[0].ptr = C.addressof(buf); s[0].off = 0; s[0].len = 5\n"
            "rc = lib.am_batch_upload(s, 1, C.byref(b))\n"
            "if rc == am.AM_ERR_NO_DEVICE: print('NODEV'); sys.exit(0)\n"
            "m = C.c_void_p(); rc = lib.am_run_batch(a.device, 0, b, C.byref(m)); print('RC', rc, lib.am_last_error())\n" % ROOT)
    out = subprocess.check_output([sys.executable, "-c", code], text=True)
    assert "NODEV" in out or ("RC %d" % am.AM_ERR_UNSUPPORTED in out and "libam_check" in out), out


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    a = am.Automaton(["abc"])               # host-side build + validation works anywhere
    for call in (lambda: a.count_matches(0, ["abc"]), lambda: a.run_text("abc"), lambda: a.run_records(0, ["abc"]),
                 lambda: am.Searcher(0, ["abc"]).contains_any("abc"), lambda: am.Replacer(0, [("a", "b")]).run("a")):
        with pytest.raises(am.AmError) as e:
            call()
        assert e.value.code == am.AM_ERR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    bad = []
    for top in ("alfred-margaret_amd", "tools", "include", os.path.join("tests", "native")):
      for base, _, files in os.walk(os.path.join(ROOT, top)):
        if os.path.basename(base) in ("lib", "__pycache__") or os.sep + "experiments" in base:      # tools/experiments: archived experiments with their own (oracle-checked) tests
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", ".sh")):
                text = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"(#include\s*[\"<][^\n]*oracle|import\s+oracle|from\s+oracle|liboracle|libam_oracle)", text):
                    bad.append(f)
    assert not bad, bad


def test_serialised_image_is_validated_before_any_device_work():
    """am_automaton_from_host_image refuses damaged blobs with AM_ERR_INVALID (header, section bounds,
    checksum) -- on a box without a GPU too, because the checks come first."""
    import numpy as np
    import torch
    from tests.helpers import ImgCheck
    lib = am.api.libam()
    img = bytes(ImgCheck().flatten(am.Automaton(["tshirt", "shirts", "shorts"]), 1))

    def load(blob):
        h = C.c_void_p()
        rc = lib.am_automaton_from_host_image(blob, len(blob), C.byref(h))
        if h.value:
            lib.am_automaton_destroy(h)
        return rc

    ok = am.AM_OK if torch.cuda.is_available() else am.AM_ERR_NO_DEVICE
    assert load(img) == ok
    assert load(img[:64]) == am.AM_ERR_INVALID                              # shorter than the header
    assert load(img[:-32]) == am.AM_ERR_INVALID                             # truncated
    assert load(b"XXXX" + img[4:]) == am.AM_ERR_INVALID                     # magic
    flipped = bytearray(img); flipped[len(img) // 2] ^= 0x40
    assert load(bytes(flipped)) == am.AM_ERR_INVALID                        # checksum
    hdr = np.frombuffer(img[:256], dtype=np.uint64).copy()
    bad = bytearray(img); bad[40:48] = np.uint64(1 << 40).tobytes()         # off_transitions far outside the blob
    assert load(bytes(bad)) == am.AM_ERR_INVALID
    assert b"image" in lib.am_last_error()


def test_switch_table_of_the_library_and_of_the_front_end_agree():
    """csrc/am_config.h holds every test / measurement switch of libam; api.DEBUG_SWITCHES is what the test suite resets after every test.  A switch known
    to only one of them would leak from one test into the next (or not be settable at all): same names, am_debug_set accepts each, refuses others."""
    src = open(os.path.join(ROOT, "alfred-margaret_amd", "csrc", "am_config.h")).read()
    names = re.findall(r'"(AM_[A-Z0-9_]+)"', src[src.index("names[kCount]"):])
    enum = re.sub(r"//[^\n]*", "", src[src.index("enum Key {"):src.index("kCount")])
    n_keys = len(re.findall(r"\bk[A-Z][A-Za-z0-9]*\b", enum))
    assert sorted(names) == sorted(am.api.DEBUG_SWITCHES) and len(set(names)) == len(names)
    assert n_keys == len(names), (n_keys, len(names))
    for n in names:
        am.debug_set(n, -1)
    with pytest.raises(am.AmError):
        am.debug_set("AM_NO_SUCH_SWITCH", 1)
