import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def st3(name):
    return np.loadtxt(os.path.join(GOLDEN, "st3", "st3_%s.dat" % name)).T


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def engine():
    """The native engine on cuda:0.  No skip, no fallback: if the HIP library or the GPU is
    missing, every gpu-marked test fails loudly."""
    from bayhunter_amd import engine as E
    return E.default_engine(0)


@pytest.fixture(autouse=True)
def _reference_search(request):
    """The GPU parity tests compare bits with the reference: every gpu-marked test starts with the process-wide engine on the
    REFERENCE's root refinement (the engine's own default is the short one, bh_engine.h; the tests of that mode select it
    themselves and `test_gpu_swd_fast.py::test_default_search_is_the_short_refinement` looks at a fresh engine) and on the
    reference's arithmetic (bh_engine_set_swd_arith; the default, the fast one, is selected by the tests of that mode)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from bayhunter_amd import engine as E
    eng = E.default_engine(0)
    before, arith = eng.swd_search(), eng.swd_arith()
    eng.set_swd_search("reference")
    eng.set_swd_arith("exact")       # (the bit-level tests of the short refinement: its restatement computes as the reference does)
    yield
    eng.set_swd_search(before)       # (what a test selected does not leak into the next one)
    eng.set_swd_arith(arith)


def rows(a, nlay):
    """[B, Lmax] padded rows -> list of trimmed 1-D arrays"""
    return [a[i, :nlay[i]] for i in range(len(nlay))]
