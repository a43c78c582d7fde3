"""The reference-shaped plugin surface on top of the engine: SurfDisp.run_model,
RFminiModRF.run_model, JointTarget.evaluate -- checked against the reference's own outputs."""
import numpy as np
import pytest

from conftest import golden, st3
import bayhunter_amd as bh

pytestmark = pytest.mark.gpu
CLS = {"rdispph": bh.RayleighDispersionPhase, "rdispgr": bh.RayleighDispersionGroup,
       "ldispph": bh.LoveDispersionPhase, "ldispgr": bh.LoveDispersionGroup}


def test_surfdisp_run_model_contract():
    g = golden("swd_golden.npz")
    for pset in ("p30", "p80"):  # p80: more than 60 periods -> linspace-60 + interp path
        per = g["x_" + pset]
        for im in (0, 3, 17, 72, 73):
            n = g["nlay"][im]
            for ir, ref in enumerate(g["refs"]):
                plugin = CLS[str(ref)](x=per, y=None).moddata.plugin
                x, y = plugin.run_model(h=g["h"][im, :n], vp=g["vp"][im, :n], vs=g["vs"][im, :n], rho=g["rho"][im, :n])
                if g["ok_" + pset][im, ir]:
                    assert np.array_equal(x, per)
                    assert np.array_equal(y, g["y_" + pset][im, ir])  # bit for bit, incl. the >60-period interp path
                else:  # in-band failure: (nan, nan), surf96_modsw.py:126
                    assert np.isnan(x) and np.isnan(y)


def test_rfmini_run_model_contract():
    g = golden("rf_golden.npz")
    tx = g["x_n201"]
    for im in (0, 4, 10):
        n = g["nlay"][im]
        for ic, (gauss, p) in enumerate(g["gauss_p"]):
            for iw, cls in enumerate((bh.PReceiverFunction, bh.SReceiverFunction)):
                plugin = cls(x=tx, y=None).moddata.plugin
                plugin.set_modelparams(gauss=gauss, p=p)
                x, y = plugin.run_model(h=g["h"][im, :n], vp=g["vp"][im, :n], vs=g["vs"][im, :n], rho=g["rho"][im, :n])
                assert np.allclose(x, tx, atol=1e-9)
                assert np.max(np.abs(y - g["y_n201"][im, ic, iw])) <= 1e-9


def test_joint_target_evaluate_matches_reference():
    g = golden("like_golden.npz")
    lawname = {"nocorr": "nocorr", "scaled": "nocorr_scalederr", "exp": "exp", "gauss": "gauss"}
    for case in g["case_names"]:
        case = str(case)
        targets = []
        for ref, law in zip(g[case + "_refs"], g[case + "_laws"]):
            ref, law = str(ref), str(law)
            if ref == "prf":
                t = bh.PReceiverFunction(g["x_rf"], g["yobs_prf"])
                t.moddata.plugin.set_modelparams(gauss=1.0, p=6.4)
            else:
                t = CLS[ref](g["x_swd"], g["yobs_" + ref], yerr=g["yerr_swd"] if law == "scaled" else None)
            t.set_noise_law(lawname[law], corr=float(g["gauss_corr"]), rcond=float(g["gauss_rcond"]))
            targets.append(t)
        jt = bh.JointTarget(targets)
        for im in range(g["nlay"].size):
            n = g["nlay"][im]
            jt.evaluate(h=g["h"][im, :n], vp=g["vp"][im, :n], vs=g["vs"][im, :n], noise=g[case + "_noise"][im])
            ref_l = g[case + "_logL"][im]
            if ref_l == -1e15:
                assert jt.proposallikelihood == -1e15 and list(jt.proposalmisfits) == [1e15] * (len(targets) + 1)
            else:
                assert abs(jt.proposallikelihood - ref_l) <= 1e-6 * abs(ref_l)
                assert np.allclose(jt.proposalmisfits, g[case + "_misfits"][im], rtol=1e-8)
                assert targets[0]._moddata_valid()


def test_user_plugin_goes_through_loglike_batch():
    """templates/myfwd.py contract: any object with run_model(h, vp, vs, rho) -> (x, y)."""
    x = np.linspace(1, 10, 12)

    class MyFwd(object):
        def run_model(self, h, vp, vs, rho, **kw):
            return x, np.full(x.size, float(np.sum(h))) + 0.1 * x

    yobs = 20.0 + 0.1 * x
    t1 = bh.SingleTarget(x, yobs, "mydata")
    t1.update_plugin(MyFwd())
    t1.set_noise_law("exp")
    t2 = bh.RayleighDispersionPhase(np.linspace(2, 40, 10), np.full(10, 3.4))
    jt = bh.JointTarget([t1, t2])
    h = np.array([5., 15., 0.]); vs = np.array([3.0, 3.6, 4.4]); vp = vs * 1.75
    jt.evaluate(h=h, vp=vp, vs=vs, noise=np.array([0.5, 0.1, 0.0, 0.05]))
    d = (20.0 + 0.1 * x) - yobs
    v = bh.Valuation()
    c_inv, ld = v.get_covariance_exp(0.5, 0.1, x.size)
    expect1 = v.get_likelihood(yobs, 20.0 + 0.1 * x, c_inv, ld)
    _, y2 = t2.moddata.plugin.run_model(h, vp, vs, vp * 0.32 + 0.77)
    c_inv, ld = v.get_covariance_nocorr(0.05, 10)
    expect2 = v.get_likelihood(t2.obsdata.y, y2, c_inv, ld)
    assert abs(jt.proposallikelihood - (expect1 + expect2)) <= 1e-9 * abs(expect1 + expect2)
    assert np.allclose(d, 0)


def test_model_params_mode_and_flsph():
    """set_modelparams(mode=, flsph=) of surf96_modsw.py:45-46 reach the device."""
    g = golden("swd_golden.npz")
    per = g["x_p30"]
    for jj, im in enumerate(g["sub_idx"][:6]):
        n = g["nlay"][im]
        for ir, ref in enumerate(g["refs"]):
            p = bh.SurfDisp(per, str(ref))
            p.set_modelparams(mode=2)
            x, y = p.run_model(g["h"][im, :n], g["vp"][im, :n], g["vs"][im, :n], g["rho"][im, :n])
            if g["ok_mode2"][jj, ir]:
                assert np.array_equal(y, g["y_mode2"][jj, ir])
            else:
                assert np.isnan(x) and np.isnan(y)


def test_synthobs_call_pattern_is_served():
    """The reference's SynthObs (src/SynthObs.py:25-99; named in the north_star, out of scope as a class) drives the
    plugins like this: Target(x=x, y=None) -> moddata.plugin.set_modelparams(mode= | gauss=, water=, p=, nsv=) ->
    run_model(h=, vp=, vs=, rho=) with keyword arguments -> (xmod, ymod).  The same calls on this package's classes
    reproduce the reference's observed-data files for the tutorial model to the files' 4-decimal rounding
    (RF: the files predate the float64 rfmini and pin it to 1e-4 absolute)."""
    h, vs, vpvs = np.array([5., 23., 8., 0.]), np.array([2.7, 3.6, 3.8, 4.4]), 1.73      # tutorial/create_testdata.py:13-15
    vp = vs * vpvs
    rho = vp * 0.32 + 0.77
    x = np.linspace(1, 41, 21)
    for ref, cls in CLS.items():
        target = cls(x=x, y=None)
        target.moddata.plugin.set_modelparams(mode=1)
        xmod, ymod = target.moddata.plugin.run_model(h=h, vp=vp, vs=vs, rho=rho)
        assert target.ref == ref and np.array_equal(xmod, x)
        assert np.max(np.abs(ymod - st3(ref)[1])) <= 6e-5
    xr = np.linspace(-5, 35, 201)
    for ref, cls in (("prf", bh.PReceiverFunction), ("srf", bh.SReceiverFunction)):
        target = cls(x=xr, y=None)
        target.moddata.plugin.set_modelparams(gauss=1.0, water=0.001, p=6.4, nsv=None)
        xmod, ymod = target.moddata.plugin.run_model(h=h, vp=vp, vs=vs, rho=rho)
        assert target.ref == ref and np.allclose(xmod, xr, atol=1e-9)
        assert np.max(np.abs(ymod - st3(ref)[1])) <= 1.5e-4
