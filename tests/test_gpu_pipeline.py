"""The two-kernel pipeline for large batches (k_filter on the caller's stream, k_consume on a high-priority side stream, slice by slice;
csrc/am_kernels.hip): forced on with AM_SF_PIPE=1 in a subprocess (the switch is read once per process), small slices,
three batch shapes; records, per-haystack counts and containsAny must equal the general kernel's on every haystack and the oracle's on a
sample.  The default bench.py run is the full-size check (its parity gate compares pipeline vs general kernel over all 10 240 haystacks)."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [{}, {"AM_SF_PIPE_SLICE_MIB": "4"}, {"AM_SF_PIPE_SLICE_MIB": "4", "AM_SF_POOL_BLOCKS": "3"}])
def test_pipeline_equals_general_kernel_and_oracle(extra):
    env = dict(os.environ, AM_SF_PIPE="1", **extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "measure", "pipeline_parity.py")], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "pipeline parity OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
