"""The CPU oracle (oracle/rf_oracle.c) against golden vectors from the reference's
RFminiModRF.run_model + compiled rfmini, and against tutorial/observed/st3_{prf,srf}.dat."""
import numpy as np
import pytest

from conftest import golden, st3


def oracle_rf(oracle, h, vp, vs, rho, tx, gauss, p, wtype):
    """what rfmini_modrf.py:99-142 does around rfmini.synrf"""
    dt = float(np.round(tx[1] - tx[0], 4))
    fsamp = 1.0 / dt
    tshft = -tx[0]
    nsamp = 2 ** int(np.ceil(np.log2(tx.size * 2)))
    z = np.concatenate(([0], np.cumsum(h)[:-1]))
    k = vp[0] / vs[0]
    poisson = (2 - k ** 2) / (2 - 2 * k ** 2)
    rf = oracle.synrf(z, vp, vs, rho, np.ones(h.size) * 500., np.ones(h.size) * 225., p, gauss, nsamp,
                      fsamp, tshft, vs[0], poisson, wtype)[2]
    return rf[:tx.size]


@pytest.mark.parametrize("axis", ["n201", "n1024"])
def test_oracle_matches_reference_golden(oracle, axis):
    g = golden("rf_golden.npz")
    tx = g["x_" + axis]
    for im in range(g["nlay"].size):
        n = g["nlay"][im]
        for ic, (gauss, p) in enumerate(g["gauss_p"]):
            for iw, w in enumerate(g["wtypes"]):
                rf = oracle_rf(oracle, g["h"][im, :n], g["vp"][im, :n], g["vs"][im, :n], g["rho"][im, :n], tx, gauss, p, str(w))
                ref = g["y_" + axis][im, ic, iw]
                assert np.max(np.abs(rf - ref)) <= 1e-10 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("ref,wtype", [("prf", "P"), ("srf", "SV")])
def test_oracle_matches_tutorial_files(oracle, ref, wtype):
    """st3_prf/srf.dat pin the RF only to ~1e-4 abs (files predate float64 rfmini)."""
    x, y = st3(ref)
    h = np.array([5., 23., 8., 0.]); vs = np.array([2.7, 3.6, 3.8, 4.4]); vp = vs * 1.73
    rf = oracle_rf(oracle, h, vp, vs, vp * 0.32 + 0.77, x, 1.0, 6.4, wtype)
    assert np.max(np.abs(rf - y)) <= 1e-4


def test_oracle_matches_compiled_reference_random(oracle):
    from oracle import refshim
    if not refshim.available():
        pytest.skip("oracle/_ref not built (needs the build container)")
    rs = np.random.RandomState(6)
    for it in range(12):
        L = rs.randint(2, 22)
        vs = np.sort(rs.uniform(2.0, 4.8, L)); h = rs.uniform(0.5, 8, L); h[-1] = 0
        vp = vs * rs.uniform(1.5, 2.0); rho = 0.32 * vp + 0.77
        z = np.concatenate(([0], np.cumsum(h)[:-1]))
        args = (z, vp, vs, rho, np.ones(L) * 500., np.ones(L) * 225., rs.uniform(4, 8), rs.uniform(1, 3), 512, 5.0, 5.0, vs[0], 0.25, "P" if it % 2 else "SV")
        assert np.array_equal(oracle.synrf(*args)[2], refshim.synrf(*args)[2])
