"""libam.so loads, exports every symbol include/am.h declares, and refuses to compute without a GPU."""
import ctypes as C
import os
import re

import pytest

import alfred_margaret_amd as am
from tests.conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "am.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(am_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 25
    lib = C.CDLL(os.path.join(am.build.LIB, "libam.so"))
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(am.api.ABI) == names      # the Python binding table covers the header exactly


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
    for base, _, files in os.walk(os.path.join(ROOT, "alfred-margaret_amd")):
        if os.path.basename(base) == "lib":
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"(#include\s*[\"<][^\n]*oracle|import\s+oracle|from\s+oracle|liboracle|libam_oracle)", text):
                    bad.append(f)
    assert not bad, bad
