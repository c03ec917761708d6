"""A small soak: 140 frames at SNRs from noise-only to clean, every candidate of every
frame through FDR + schedule on the GPU vs the oracle, bit for bit (tools/soak_parity.py
runs the same check on thousands of frames)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("n,hbw,maxdrift", [(140, 10, 0), (42, 60, 0), (84, 10, 3)])
def test_soak(n, hbw, maxdrift):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_parity.py"), str(n), str(hbw),
                        str(maxdrift)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout


@pytest.mark.gpu
def test_recording_snr_sweep_gpu_equals_cpu():
    """BASELINE configs[4] in small: the reference's example recording + AWGN at
    -20..-30 dB, 3 noise seeds each, decoded end to end on the GPU and by the CPU
    path (oracle + host tail) on the same frames: identical decode sets, and every
    record down to -26 dB yields `VE3EMB FN42 33`."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "snr_sweep.py"), "3"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout
    for snr in ("-20.0", "-22.0", "-24.0", "-26.0"):
        assert ("SNR %s dB:  3/3 decoded" % snr) in r.stdout, r.stdout
