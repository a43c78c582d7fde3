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


def rows(a, nlay):
    """[B, Lmax] padded rows -> list of trimmed 1-D arrays"""
    return [a[i, :nlay[i]] for i in range(len(nlay))]
