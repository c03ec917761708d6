import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build(ref=os.path.isdir("/root/reference/lib"))
    return oracle_py


@pytest.fixture(scope="session")
def G():
    import gr_uwspr_amd
    return gr_uwspr_amd


@pytest.fixture(scope="session")
def ve3emb(oracle):
    return oracle.read_c2(os.path.join(GOLDEN, "VE3EMB.c2"))
