"""A small soak: 140 frames at SNRs from noise-only to clean, every candidate of every
frame through FDR + schedule on the GPU vs the oracle, bit for bit (tools/soak_parity.py
runs the same check on thousands of frames)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("n,hbw", [(140, 10), (42, 60)])
def test_soak(n, hbw):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_parity.py"), str(n), str(hbw)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout
