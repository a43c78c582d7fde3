"""Oracle forward models + dense likelihood (oracle/like_oracle.c) against
JointTarget.evaluate of the reference (tests/golden/like_golden.npz)."""
import numpy as np
import pytest

from conftest import golden
from test_oracle_rf import oracle_rf
from test_oracle_swd import REFS, run

LAWMAP = {"nocorr": 0, "scaled": 1, "exp": 2, "gauss": 3}


def gauss_rinv(corr, n, rcond):
    idx = np.arange(n)
    R = corr ** ((idx[:, None] - idx[None, :]).astype(float) ** 2)
    return np.linalg.pinv(R, rcond=rcond), np.linalg.slogdet(R)[1]


def oracle_joint(oracle, g, case, im):
    """(logL, misfits) the way Targets.py:314-347 forms them, with the oracle's pieces."""
    n = g["nlay"][im]
    h, vp, vs = g["h"][im, :n], g["vp"][im, :n], g["vs"][im, :n]
    rho = vp * 0.32 + 0.77
    refs, laws = g[case + "_refs"], g[case + "_laws"]
    noise = g[case + "_noise"][im]
    logL, misfits = 0.0, []
    for it, (ref, law) in enumerate(zip(refs, laws)):
        ref, law = str(ref), str(law)
        corr, sigma = noise[2 * it:2 * it + 2]
        if ref == "prf":
            x = g["x_rf"]
            ymod = oracle_rf(oracle, h, vp, vs, rho, x, 1.0, 6.4, "P")
        else:
            x = g["x_swd"]
            err, ymod = run(oracle, h, vp, vs, rho, x, *REFS[ref])
            if err:
                return -1e15, [1e15] * (len(refs) + 1)
        yobs = g["yobs_" + ref]
        kw = {}
        if law == "scaled":
            kw["yerr"] = g["yerr_swd"]
        if law == "gauss":
            kw["rinv"], kw["logdet_r"] = gauss_rinv(float(g["gauss_corr"]), x.size, float(g["gauss_rcond"]))
        logL += oracle.loglike_dense(LAWMAP[law], ymod, yobs, corr, sigma, **kw)
        misfits.append(oracle.rms(ymod, yobs))
    return logL, misfits + [sum(misfits)]


@pytest.mark.parametrize("case", ["swd_nocorr", "swd_scaled", "swd_exp", "joint_exp", "joint_gauss"])
def test_oracle_likelihood_matches_reference(oracle, case):
    g = golden("like_golden.npz")
    for im in range(g["nlay"].size):
        logL, misfits = oracle_joint(oracle, g, case, im)
        ref = g[case + "_logL"][im]
        # the Gauss law goes through a pseudo-inverse of an ill-conditioned R (rcond 1e-6):
        # summation order matters at the 1e-9 level there
        tol = 1e-7 if case == "joint_gauss" else 1e-10
        assert abs(logL - ref) <= tol * abs(ref), (case, im, logL, ref)
        assert np.allclose(misfits, g[case + "_misfits"][im], rtol=1e-10, atol=0)
