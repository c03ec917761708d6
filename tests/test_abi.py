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


def test_build_is_shared_by_ranks_and_independent_of_the_tree_path(G, tmp_path):
    """What the driver's multi-GPU launch does: N ranks import the package at the same moment, from whatever path the
    tree was copied to.  The library that travelled with the tree must be taken as it is -- no rank recompiles it
    (the stamp holds flags + source contents, not paths or file times), and concurrent callers neither race for the
    output file nor block each other for longer than the check."""
    import subprocess
    import sys
    N = G.native
    N.build()
    before = (os.path.getmtime(N.LIBPATH), os.path.getsize(N.LIBPATH))
    repo = os.path.abspath(os.path.join(os.path.dirname(N.CSRC), ".."))
    other = tmp_path / "another_root"
    os.symlink(repo, other)                               # the same tree under another absolute path
    code = ("import sys, time; sys.path.insert(0, %r); t = time.time(); import gr_uwspr_amd as G; G.build(); "
            "L = G.native.lib(); assert hasattr(L, 'uwspr_ctx_create'); print('%%.1f' %% (time.time() - t))" % str(other))
    procs = [subprocess.Popen([sys.executable, "-c", code], cwd=str(other), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(4)]
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err[-2000:]
        assert float(out.strip().splitlines()[-1]) < 60.0, out       # (an import, not a compile)
    assert (os.path.getmtime(N.LIBPATH), os.path.getsize(N.LIBPATH)) == before
