import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _checker_kernel(request):
    """The general AC-walk kernel k_ac -- the tests' independent second algorithm, Automaton.set_kernel(1) -- is test infrastructure: it lives
    in libam_check.so (tests/native/am_ac.hip), not in the product library, and registers itself with libam when loaded.  Only the GPU tests run it: they get it
    loaded (once); the CPU suites (the oracle, the host mirror, the host interpreter of the image) run on a box without hipcc / the HIP runtime as they are."""
    if request.node.get_closest_marker("gpu") is not None:
        import alfred_margaret_amd as am
        am.api.load_check()
    yield


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return json.load(f)


CASES = {"CaseSensitive": 0, "IgnoreCase": 1}


@pytest.fixture(autouse=True)
def _reset_debug_switches():
    """Tests flip libam's test / measurement switches with am.debug_set (csrc/am_config.h); none may leak into the next test."""
    yield
    import alfred_margaret_amd as am
    if am.api._libam is not None:
        am.debug_reset()


def pytest_sessionfinish(session, exitstatus):
    """tools/bounds_check.sh: a -DAM_BOUNDS_CHECK build of libam counts the index assertions its kernels failed (csrc/am_bounds.h); any count fails the session."""
    if not os.environ.get("AM_BOUNDS_CHECK"):
        return
    import alfred_margaret_amd as am
    if am.api._libam is None:
        return
    failed, line, units = am.api.bounds_report()
    print("\n[bounds] %d translation units carry index assertions; %d assertions failed%s" % (units, failed, "" if not failed else " -- " + (am.api.libam().am_last_error() or b"").decode()))
    if failed or units == 0:
        session.exitstatus = 1
