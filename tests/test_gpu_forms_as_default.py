"""Every fixture-based comparison of tests/test_gpu_parity.py -- golden vectors, the oracle record by record, frame edges,
drifting and hand-made candidates, lazy + resume, the stream ring -- again with ANOTHER schedule form as the process-wide
default (UWSPR_OPTIONS, the one variable the library reads): the staged launches bench.py times, and the flat kernel.
Round 5's review: those comparisons reached the staged form only through staged == fused; here they reach it directly."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("options", ["sched=0", "sched=0,stage_kernels=0"])
def test_parity_suite_with_another_default_form(options):
    env = dict(os.environ, UWSPR_OPTIONS=options)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = r.stdout[-1500:] + r.stderr[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
