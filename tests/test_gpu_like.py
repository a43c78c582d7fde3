"""Fused forward model + likelihood (bh_evaluate_batch / bh_loglike_batch) against the
reference's JointTarget.evaluate (golden) and the oracle's dense formulation."""
import numpy as np
import pytest

from conftest import golden
from bayhunter_amd import engine as E
from test_oracle_like import LAWMAP, gauss_rinv, oracle_joint

pytestmark = pytest.mark.gpu
REFS = {"rdispph": (2, 0), "rdispgr": (2, 1), "ldispph": (1, 0), "ldispgr": (1, 1)}


def descs_for(g, case):
    out = []
    for ref, law in zip(g[case + "_refs"], g[case + "_laws"]):
        ref, law = str(ref), str(law)
        d = {"law": LAWMAP[law], "yobs": g["yobs_" + ref]}
        if ref == "prf":
            x = g["x_rf"]
            d.update(kind=E.TARGET_RF, n=x.size, waveno=0, nsamp=512, p=6.4, gauss=1.0, fsamp=5.0, tshift=5.0)
        else:
            x = g["x_swd"]
            d.update(kind=E.TARGET_SWD, n=x.size, x=x, iwave=REFS[ref][0], igr=REFS[ref][1])
        if law == "scaled":
            d["yerr"] = g["yerr_swd"]
        if law == "gauss":
            d["rinv"], d["logdet_r"] = gauss_rinv(float(g["gauss_corr"]), x.size, float(g["gauss_rcond"]))
        out.append(d)
    return out


@pytest.mark.parametrize("case", ["swd_nocorr", "swd_scaled", "swd_exp", "joint_exp", "joint_gauss"])
def test_evaluate_matches_reference_and_oracle(engine, oracle, case):
    g = golden("like_golden.npz")
    engine.set_targets(descs_for(g, case))
    noise = g[case + "_noise"]
    logL, misf, err = engine.evaluate_batch(g["nlay"], g["h"], g["vp"], g["vs"], noise, layout="model_major")
    ref_logL, ref_misf = g[case + "_logL"], g[case + "_misfits"]
    failed = ref_logL == -1e15
    assert np.array_equal(err != 0, failed)
    assert np.all(logL[failed] == -1e15) and np.all(misf[failed] == 1e15)
    tol = 1e-6 if case == "joint_gauss" else 1e-8   # aim <= 1e-8 relative
    relerr = np.abs(logL - ref_logL) / np.abs(ref_logL)
    assert np.all(relerr[~failed] <= tol)
    assert np.allclose(misf[~failed], ref_misf[~failed], rtol=1e-8, atol=0)
    for im in range(g["nlay"].size):
        o_logL, _ = oracle_joint(oracle, g, case, im)
        assert abs(logL[im] - o_logL) <= tol * abs(o_logL)


def test_rho_argument_and_ymod_output(engine):
    g = golden("like_golden.npz")
    engine.set_targets(descs_for(g, "joint_exp"))
    noise = g["joint_exp_noise"]
    a = engine.evaluate_batch(g["nlay"], g["h"], g["vp"], g["vs"], noise, layout="model_major")
    b = engine.evaluate_batch(g["nlay"], g["h"], g["vp"], g["vs"], noise, rho=g["vp"] * 0.32 + 0.77,
                              layout="model_major", want_ymod=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ymod = b[3]
    assert ymod.shape == (g["nlay"].size, 30 + 30 + 201)
    # likelihood-only entry point on the same synthetics gives the same numbers
    fail = np.zeros((3, g["nlay"].size), dtype=np.int32); fail[:, a[2] != 0] = 1
    c = engine.loglike_batch(ymod, noise, fail)
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])


def test_exponential_law_edge_sizes(engine, oracle):
    """n = 1 and n = 2: get_corr_inv's d[0] = d[-1] = 1 (Targets.py:131-137)."""
    rs = np.random.RandomState(4)
    for n in (1, 2, 3):
        yobs = rs.normal(0, 1, n)
        engine.set_targets([{"kind": E.TARGET_USER, "law": E.LAW_EXP, "n": n, "yobs": yobs}])
        ymod = yobs + rs.normal(0, 0.1, (5, n))
        noise = np.column_stack((rs.uniform(0.2, 0.8, 5), rs.uniform(0.01, 0.1, 5)))
        logL, misf, err = engine.loglike_batch(ymod, noise)
        for b in range(5):
            if n > 1:
                o = oracle.loglike_dense(2, ymod[b], yobs, noise[b, 0], noise[b, 1])
            else:  # dense n=1: c_inv = 1/(s^2 (1-r^2))
                r, s = noise[b]
                o = -0.5 * (np.log(2 * np.pi) + 2 * np.log(s)) - 0.5 * (ymod[b, 0] - yobs[0]) ** 2 / (s * s * (1 - r * r))
            assert abs(logL[b] - o) <= 1e-10 * abs(o)
            assert abs(misf[b, 0] - oracle.rms(ymod[b], yobs)) <= 1e-12


@pytest.mark.parametrize("n,B", [(1024, 300), (201, 4096), (70, 65), (16, 3)])
def test_gauss_law_mfma_contraction(engine, n, B):
    """d^T R^-1 d through the FP64-MFMA kernel (csrc/gauss_kernel.hip) against NumPy's dense form
    (Targets.py:162-173, :339-342), including sizes that are not multiples of the 64x64 tiles."""
    rs = np.random.RandomState(n)
    idx = np.arange(n)
    R = 0.9 ** ((idx[:, None] - idx[None, :]).astype(float) ** 2)
    rinv = np.linalg.pinv(R, rcond=1e-6)
    ld = np.linalg.slogdet(R)[1]
    yobs = rs.normal(0, 0.1, n)
    engine.set_targets([{"kind": E.TARGET_USER, "law": E.LAW_GAUSS, "n": n, "yobs": yobs, "rinv": rinv, "logdet_r": ld}])
    ymod = yobs + rs.normal(0, 0.01, (B, n))
    noise = np.column_stack((np.full(B, 0.9), rs.uniform(0.005, 0.05, B)))
    logL, misf, err = engine.loglike_batch(ymod, noise)
    d = ymod - yobs
    phi = np.einsum("bi,ij,bj->b", d, rinv, d)
    ref = -0.5 * (n * np.log(2 * np.pi) + 2 * n * np.log(noise[:, 1]) + ld) - 0.5 * phi / noise[:, 1] ** 2
    assert np.max(np.abs(logL - ref) / np.abs(ref)) <= 1e-10
    assert np.allclose(misf[:, 0], np.sqrt(np.mean(d * d, axis=1)), rtol=1e-12)
    again = engine.loglike_batch(ymod, noise)[0]
    assert np.array_equal(again, logL)  # fixed summation order: deterministic


def test_fused_target_with_more_than_60_periods(engine):
    """n > 60 in a fused SWD target: forward model on linspace(min, max, 60), np.interp back
    (surf96_modsw.py:35-43, :119-122) -- against the reference's own run_model outputs."""
    g = golden("swd_golden.npz")
    per = g["x_p80"]
    yobs = np.full(per.size, 3.0)
    for ir, ref in enumerate(g["refs"]):
        iwave, igr = {"rdispph": (2, 0), "rdispgr": (2, 1), "ldispph": (1, 0), "ldispgr": (1, 1)}[str(ref)]
        engine.set_targets([{"kind": E.TARGET_SWD, "law": E.LAW_NOCORR, "n": per.size, "x": per, "yobs": yobs,
                             "iwave": iwave, "igr": igr}])
        B = g["nlay"].size
        noise = np.tile([0.0, 0.05], (B, 1))
        logL, misf, err, ymod = engine.evaluate_batch(g["nlay"], g["h"], g["vp"], g["vs"], noise, rho=g["rho"],
                                                      layout="model_major", want_ymod=True)
        ok = g["ok_p80"][:, ir].astype(bool)
        assert np.array_equal(err == 0, ok)
        assert np.array_equal(ymod[ok], g["y_p80"][:, ir][ok])       # bit-identical, np.interp included
        d = g["y_p80"][:, ir][ok] - yobs
        ref = -0.5 * (per.size * np.log(2 * np.pi) + 2 * per.size * np.log(0.05)) - 0.5 * np.sum(d * d, axis=1) / 0.05 ** 2
        assert np.max(np.abs(logL[ok] - ref) / np.abs(ref)) <= 1e-12


@pytest.mark.parametrize("law", [E.LAW_NOCORR, E.LAW_EXP])
@pytest.mark.parametrize("n,nsamp,fsamp", [(1024, 2048, 20.0), (201, 512, 5.0), (48, 128, 4.0), (1, 128, 4.0), (2, 128, 4.0), (8192, 16384, 50.0),
                                           (16385, 32768, 100.0)])   # (beyond 16 384 samples the fused sums read the spectrum from the HBM workspace)
def test_fused_receiver_function_likelihood_has_the_bits_of_the_unfused_one(engine, law, n, nsamp, fsamp):
    """bh_evaluate_batch without synthetics asked for: the receiver function's samples never leave the CU -- the synthesis
    kernel forms the sums the nocorr / exponential law needs in like_kernel's own order (RfKernelArgs::sums).  logL and
    misfits are bit-identical to the call that writes the trace and reads it back (synthetics asked for), for traces up to
    64 samples (one wavefront's reduction), longer ones, and the 1- and 2-sample edge cases of the exponential law."""
    from bayhunter_amd.synth import synth_models
    rs = np.random.RandomState(n)
    B = 96 if nsamp <= 16384 else 12
    nlay, h, vp, vs, rho = synth_models(rs, B, 10, ragged=True)
    per = np.linspace(2, 40, 20)
    engine.set_targets([
        {"kind": E.TARGET_SWD, "law": E.LAW_NOCORR, "n": per.size, "x": per, "yobs": 3.4 + 0.01 * per, "iwave": 2, "igr": 0},
        {"kind": E.TARGET_RF, "law": law, "n": n, "yobs": rs.normal(0, 0.05, n), "waveno": 0, "nsamp": nsamp, "p": 6.4,
         "gauss": 2.0, "fsamp": fsamp, "tshift": 5.0}])
    noise = np.column_stack([np.zeros(B), rs.uniform(0.01, 0.05, B), rs.uniform(0.3, 0.8, B), rs.uniform(0.005, 0.05, B)])
    L1, m1, e1 = engine.evaluate_batch(nlay, h, vp, vs, noise)                         # fused
    L2, m2, e2, y2 = engine.evaluate_batch(nlay, h, vp, vs, noise, want_ymod=True)     # the trace written and read back
    assert np.array_equal(e1, e2) and np.array_equal(L1, L2) and np.array_equal(m1, m2)
    assert np.all(np.isfinite(L1[e1 == 0]))
