"""include/am.h consumed from plain C (gcc, no C++ runtime of our own, no Python in the loop): tests/c/abi_driver.c is
compiled against the header, linked with libam.so and run as a separate process.  Without a GPU the link must succeed and
the driver must report AM_ERR_NO_DEVICE (exit code 2); on the MI355X its output must equal the oracle's."""
import os
import subprocess

import numpy as np
import pytest

import alfred_margaret_amd as am
from oracle import oracle
from tests.conftest import ROOT
from tests.helpers import expand_records, oracle_triples


def _build(tmp_path):
    exe = str(tmp_path / "abi_driver")
    lib = am.build.LIB
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_driver.c"),
                           "-L", lib, "-lam", "-Wl,-rpath," + lib, "-o", exe])
    return exe


def _dump(tmp_path, needles, hay):
    m = oracle.Machine(needles)                      # packed arrays exactly as the reference's build makes them
    m.transitions().tofile(tmp_path / "tr"); m.offsets().tofile(tmp_path / "of"); m.root_ascii().tofile(tmp_path / "ra")
    np.diff(m.values_off()).astype(np.uint32).tofile(tmp_path / "vl")
    (tmp_path / "hay").write_bytes(hay)
    return m, [str(tmp_path / n) for n in ("tr", "of", "ra", "vl", "hay")]


def test_c_driver_links_and_reports_no_device(tmp_path):
    import torch
    am.api.libam()                                    # make sure the library is built
    exe = _build(tmp_path)
    _, files = _dump(tmp_path, ["tshirt", "shirts", "shorts"], b"short tshirts")
    p = subprocess.run([exe] + files + ["0"], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert p.returncode == 0, p.stderr
    else:
        assert p.returncode == 2 and "no HIP device" in p.stderr, (p.returncode, p.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [0, 1])
def test_c_driver_matches_the_oracle(tmp_path, case):
    exe = _build(tmp_path)
    needles = ["tshirt", "shirts", "shorts", "ß", "k", "sweat"]
    hay = ("short tshirts and SWEATSHIRTS, Kelvin K ẞ ß " * 40).encode("utf-8")
    m, files = _dump(tmp_path, needles, hay)
    p = subprocess.run([exe] + files + [str(case)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.split("\n")
    assert lines[0] == "count %d" % m.count_matches(case, hay)
    assert lines[1] == "any %d" % int(m.contains_any(case, hay))
    k = int(lines[2].split()[1])
    recs = [tuple(int(x) for x in l.split()) for l in lines[3:3 + k]]
    got = expand_records(m.values_off(), m.values(), [0] * k, [s for _, s in recs], [e for e, _ in recs])
    assert got == oracle_triples(m, case, [hay]) and k > 50
