"""CPU-side checks of the drop-in boundary: the shared library builds for
gfx950, loads, exports every symbol include/uwspr_hip.h declares, mirrors the
candidate_t layout, and FAILS LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(G):
    N = G.native
    N.build()
    L = N.lib()
    hdr = open(os.path.join(os.path.dirname(N.CSRC), "..", "include", "uwspr_hip.h")).read()
    declared = set(re.findall(r"\b(uwspr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(N.ABI_SYMBOLS), declared ^ set(N.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_record_layouts(G):
    N = G.native
    assert N.CAND_DTYPE.itemsize == 48            # candidate_t, lib/candidate_t.h:27-50
    assert N.CAND_DTYPE.fields["shift"][1] == 16 and N.CAND_DTYPE.fields["V1"][1] == 24
    assert N.CAND_DTYPE.fields["p2"][1] == 44
    assert N.HYP_DTYPE.itemsize == 48 and N.DEMOD_DTYPE.itemsize == 2980


def test_parameter_errors_come_before_any_device_use(G):
    """FDR_impl.cc:85-90 exit(-1)s on halfbandwidth > fs/2; the GRC default 187 makes
    the reference read out of bounds.  Both are status codes here, GPU or not."""
    N = G.native
    with pytest.raises(N.UwsprError) as e:
        G.Context(halfbandwidth=200)
    assert e.value.status == -1 and "Half pass bandwidth" in str(e.value)
    with pytest.raises(N.UwsprError) as e:
        G.Context(halfbandwidth=187)
    assert e.value.status == -2
    with pytest.raises(N.UwsprError) as e:
        G.Context(spb=128)
    assert e.value.status == -3


def test_no_gpu_means_loud_failure_not_a_fallback(G):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    N = G.native
    with pytest.raises(N.UwsprError) as e:
        G.Context()
    assert e.value.status == -7 and "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for base in ("gr-uwspr_amd", "include"):
        for dp, _, files in os.walk(os.path.join(root, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".cc")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"oracle_py|uwspr_oracle|liboracle|libuwspr_ref", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
