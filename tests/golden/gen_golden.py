#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference and `make -C oracle ref`).  It
imports the reference's own Python layer (Targets, Models, SynthObs, surf96_modsw,
rfmini_modrf) unmodified from /root/reference/src, with
  * `BayHunter.surfdisp96_ext` / `BayHunter.rfmini` provided by oracle/refshim.py, i.e. the
    unmodified Fortran / C++ sources compiled to oracle/_ref/*.so,
  * stub `zmq` / `configobj` modules (not installed here, not on the hot path),
  * the NumPy aliases the reference still uses (np.float, np.int, np.product).
Nothing of the reference is copied: the outputs are data (inputs + expected outputs).

    python tests/golden/gen_golden.py          # rewrites tests/golden/*.npz and st3/*.dat
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("BH_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


def import_reference():
    from oracle import refshim
    assert refshim.available(), "run `make -C oracle ref` first"
    scratch = tempfile.mkdtemp(prefix="bhref_")
    os.symlink(os.path.join(REF, "src"), os.path.join(scratch, "BayHunter"))
    sys.path.insert(0, scratch)
    np.float = float
    np.int = int
    np.product = np.prod
    zmq = types.ModuleType("zmq")
    zmq.Socket = type("Socket", (), {})
    zmq.Context = type("Context", (), {})
    zmq.PUB, zmq.SUB, zmq.SNDMORE = 1, 2, 2
    sys.modules["zmq"] = zmq
    cfg = types.ModuleType("configobj")

    class ConfigObj(dict):  # just enough of configobj for utils.load_params on defaults.ini
        def __init__(self, path):
            dict.__init__(self)
            self.sections = []
            cur = None
            for line in open(path):
                line = line.split("#")[0].strip()
                if not line:
                    continue
                if line.startswith("["):
                    cur = line.strip("[]")
                    self[cur] = {}
                    self.sections.append(cur)
                else:
                    k, v = [t.strip() for t in line.split("=", 1)]
                    # configobj splits on top-level commas only
                    parts, depth, tok = [], 0, ""
                    for ch in v:
                        if ch in "([":
                            depth += 1
                        if ch in ")]":
                            depth -= 1
                        if ch == "," and depth == 0:
                            parts.append(tok.strip()); tok = ""
                        else:
                            tok += ch
                    parts.append(tok.strip())
                    self[cur][k] = parts if len(parts) > 1 else parts[0]

    cfg.ConfigObj = ConfigObj
    sys.modules["configobj"] = cfg
    import matplotlib
    matplotlib.use("Agg")
    ext = types.ModuleType("BayHunter.surfdisp96_ext")
    ext.surfdisp96 = refshim.surfdisp96
    sys.modules["BayHunter.surfdisp96_ext"] = ext
    rfm = types.ModuleType("BayHunter.rfmini")
    rfm.synrf = refshim.synrf
    sys.modules["BayHunter.rfmini"] = rfm
    import BayHunter  # noqa: F401  (the reference package itself)
    Targets = sys.modules["BayHunter.Targets"]
    Models = sys.modules["BayHunter.Models"]
    return Targets, Models, sys.modules["BayHunter.SynthObs"].SynthObs


def random_model(rs, L, kind):
    """kind: 0 monotone, 1 one low-velocity layer, 2 thin layers, 3 vp/vs extremes"""
    vs = np.sort(rs.uniform(2.0, 4.8, L))
    h = rs.uniform(1.5, 8.0, L)
    vpvs = rs.uniform(1.6, 1.9)
    if kind == 1 and L > 3:
        i = rs.randint(1, L - 1)
        vs[i] = 0.9 * vs[i - 1]
    if kind == 2:
        h = rs.uniform(0.1, 1.0, L)
    if kind == 3:
        vpvs = rs.choice([1.4, 1.45, 2.05, 2.1])
    h[-1] = 0.0
    vp = vs * vpvs
    rho = 0.32 * vp + 0.77
    return h, vp, vs, rho


def pad(rows, Lmax):
    out = np.zeros((len(rows), Lmax))
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out


def main():
    Targets, Models, SynthObs = import_reference()
    rs = np.random.RandomState(20260927)
    Lmax = 21

    # ---- 0. the reference's own golden files (data) --------------------------------------
    st3 = os.path.join(HERE, "st3")
    os.makedirs(st3, exist_ok=True)
    for f in sorted(os.listdir(os.path.join(REF, "tutorial", "observed"))):
        shutil.copyfile(os.path.join(REF, "tutorial", "observed", f), os.path.join(st3, f))

    # ---- 1. surface-wave dispersion through SurfDisp.run_model ----------------------------
    refs = ["rdispph", "rdispgr", "ldispph", "ldispgr"]
    cls = {"rdispph": Targets.RayleighDispersionPhase, "rdispgr": Targets.RayleighDispersionGroup,
           "ldispph": Targets.LoveDispersionPhase, "ldispgr": Targets.LoveDispersionGroup}
    period_sets = {"p21": np.linspace(1, 41, 21), "p30": np.linspace(2, 60, 30),
                   "p80": np.linspace(1.5, 70, 80)}  # p80 exercises the >60 resampling path
    models = []
    for i in range(72):
        L = int(rs.randint(2, Lmax + 1))
        models.append(random_model(rs, L, i % 4))
    # plus the tutorial model and a model the reference fails on (strong inversion)
    h = np.array([5., 23., 8., 0.]); vs = np.array([2.7, 3.6, 3.8, 4.4]); vp = vs * 1.73
    models.append((h, vp, vs, vp * 0.32 + 0.77))
    h = np.array([2., 3., 10., 0.]); vs = np.array([3.9, 1.6, 4.4, 3.0]); vp = vs * 1.75
    models.append((h, vp, vs, vp * 0.32 + 0.77))
    out = {"nlay": np.array([len(m[0]) for m in models], dtype=np.int32),
           "h": pad([m[0] for m in models], Lmax), "vp": pad([m[1] for m in models], Lmax),
           "vs": pad([m[2] for m in models], Lmax), "rho": pad([m[3] for m in models], Lmax),
           "refs": np.array(refs)}
    for pname, per in period_sets.items():
        y = np.zeros((len(models), 4, per.size))
        ok = np.zeros((len(models), 4), dtype=np.int32)
        for im, (h, vp, vs, rho) in enumerate(models):
            for ir, ref in enumerate(refs):
                plugin = cls[ref](x=per, y=None).moddata.plugin
                xm, ym = plugin.run_model(h=h, vp=vp, vs=vs, rho=rho)
                if isinstance(ym, np.ndarray):
                    assert np.array_equal(xm, per)
                    y[im, ir] = ym
                    ok[im, ir] = 1
                else:
                    y[im, ir] = np.nan
        out["x_" + pname] = per
        out["y_" + pname] = y
        out["ok_" + pname] = ok
        print("swd", pname, "valid", ok.sum(), "of", ok.size)
    # higher modes / spherical flag on a sub-set (p30 only)
    per = period_sets["p30"]
    sub = list(range(0, 24))
    for tag, kw in (("mode2", dict(mode=2)), ("sph", dict(flsph=1))):
        y = np.zeros((len(sub), 4, per.size))
        ok = np.zeros((len(sub), 4), dtype=np.int32)
        for jj, im in enumerate(sub):
            h, vp, vs, rho = models[im]
            for ir, ref in enumerate(refs):
                plugin = cls[ref](x=per, y=None).moddata.plugin
                plugin.set_modelparams(**kw)
                xm, ym = plugin.run_model(h=h, vp=vp, vs=vs, rho=rho)
                if isinstance(ym, np.ndarray):
                    y[jj, ir] = ym
                    ok[jj, ir] = 1
                else:
                    y[jj, ir] = np.nan
        out["y_" + tag] = y
        out["ok_" + tag] = ok
    out["sub_idx"] = np.array(sub, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "swd_golden.npz"), **out)

    # ---- 2. receiver functions through RFminiModRF.run_model -------------------------------
    rfmodels = [models[i] for i in (0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 72)]  # 72 = st3
    time_axes = {"n201": np.linspace(-5, 35, 201), "n1024": -5 + 0.05 * np.arange(1024)}
    combos = [(1.0, 6.4), (2.5, 6.4), (2.5, 4.0), (1.0, 8.0)]
    out = {"nlay": np.array([len(m[0]) for m in rfmodels], dtype=np.int32),
           "h": pad([m[0] for m in rfmodels], Lmax), "vp": pad([m[1] for m in rfmodels], Lmax),
           "vs": pad([m[2] for m in rfmodels], Lmax), "rho": pad([m[3] for m in rfmodels], Lmax),
           "gauss_p": np.array(combos), "wtypes": np.array(["P", "SV"])}
    for tname, tx in time_axes.items():
        y = np.zeros((len(rfmodels), len(combos), 2, tx.size))
        for im, (h, vp, vs, rho) in enumerate(rfmodels):
            for ic, (g, p) in enumerate(combos):
                for iw, c in enumerate((Targets.PReceiverFunction, Targets.SReceiverFunction)):
                    plugin = c(x=tx, y=None).moddata.plugin
                    plugin.set_modelparams(gauss=g, p=p)
                    xm, ym = plugin.run_model(h=h, vp=vp, vs=vs, rho=rho)
                    assert xm.size == tx.size and np.allclose(xm, tx, atol=1e-9)
                    y[im, ic, iw] = ym
        out["x_" + tname] = tx
        out["y_" + tname] = y
        print("rf", tname, "peak", np.abs(y).max(), "finite", np.isfinite(y).all())
    np.savez_compressed(os.path.join(HERE, "rf_golden.npz"), **out)

    # ---- 3. JointTarget.evaluate: logL + misfits for the four covariance laws ---------------
    per = period_sets["p30"]
    tx = time_axes["n201"]
    truth = models[72]
    sw = SynthObs.return_swddata(truth[0], truth[2], vpvs=1.73, x=per)
    rf = SynthObs.return_rfdata(truth[0], truth[2], vpvs=1.73, x=tx, pars=dict(gauss=1.0, p=6.4))
    nrs = np.random.RandomState(7)
    yobs = {r: sw[r][1] + nrs.normal(0, 0.012, per.size) for r in refs}
    yobs["prf"] = rf["prf"][1] + nrs.normal(0, 0.005, tx.size)
    yerr_sw = nrs.uniform(0.01, 0.05, per.size)
    cases = []  # (name, [(ref, law, corr_fixed_or_None)], with_yerr)
    cases.append(("swd_nocorr", [("rdispph", "nocorr"), ("ldispph", "nocorr")]))
    cases.append(("swd_scaled", [("rdispph", "scaled"), ("rdispgr", "scaled")]))
    cases.append(("swd_exp", [("rdispph", "exp"), ("ldispph", "exp"), ("ldispgr", "exp")]))
    cases.append(("joint_exp", [("rdispph", "nocorr"), ("ldispph", "nocorr"), ("prf", "exp")]))
    cases.append(("joint_gauss", [("rdispph", "nocorr"), ("prf", "gauss")]))
    evalmodels = [models[i] for i in (72, 0, 4, 8, 12, 16, 20, 73)]  # 73 fails in surf96
    out = {"nlay": np.array([len(m[0]) for m in evalmodels], dtype=np.int32),
           "h": pad([m[0] for m in evalmodels], Lmax), "vp": pad([m[1] for m in evalmodels], Lmax),
           "vs": pad([m[2] for m in evalmodels], Lmax), "x_swd": per, "x_rf": tx,
           "yerr_swd": yerr_sw, "gauss_corr": 0.92, "gauss_rcond": 1e-6,
           "case_names": np.array([c[0] for c in cases])}
    for r in yobs:
        out["yobs_" + r] = yobs[r]
    for name, spec in cases:
        targets = []
        for ref, law in spec:
            x = tx if ref == "prf" else per
            ye = yerr_sw if law == "scaled" else None
            if ref == "prf":
                t = Targets.PReceiverFunction(x=x, y=yobs[ref])
                t.moddata.plugin.set_modelparams(gauss=1.0, p=6.4)
            else:
                t = cls[ref](x=x, y=yobs[ref], yerr=ye)
            # install the covariance law exactly the way SingleChain.set_target_covariance does
            if law == "nocorr":
                t.get_covariance = t.valuation.get_covariance_nocorr
            elif law == "scaled":
                t.get_covariance = t.valuation.get_covariance_nocorr_scalederr
            elif law == "exp":
                t.get_covariance = t.valuation.get_covariance_exp
            elif law == "gauss":
                t.valuation.init_covariance_gauss(0.92, x.size, rcond=1e-6)
                t.get_covariance = t.valuation.get_covariance_gauss
            targets.append(t)
        jt = Targets.JointTarget(targets=targets)
        nt = len(targets)
        noise = np.zeros((len(evalmodels), 2 * nt))
        logl = np.zeros(len(evalmodels))
        misf = np.zeros((len(evalmodels), nt + 1))
        for im, (h, vp, vs, rho) in enumerate(evalmodels):
            for it, (ref, law) in enumerate(spec):
                corr = 0.0 if law in ("nocorr", "scaled") else (0.92 if law == "gauss" else nrs.uniform(0.35, 0.75))
                sigma = nrs.uniform(0.005, 0.05)
                noise[im, 2 * it:2 * it + 2] = (corr, sigma)
            jt.evaluate(h=h, vp=vp, vs=vs, noise=noise[im])
            logl[im] = jt.proposallikelihood
            misf[im] = np.asarray(jt.proposalmisfits, dtype=float)
        out[name + "_refs"] = np.array([s[0] for s in spec])
        out[name + "_laws"] = np.array([s[1] for s in spec])
        out[name + "_noise"] = noise
        out[name + "_logL"] = logl
        out[name + "_misfits"] = misf
        print("like", name, logl)
    np.savez_compressed(os.path.join(HERE, "like_golden.npz"), **out)

    # ---- 4. Model.get_vp_vs_h --------------------------------------------------------------
    vec, vpvs_l, mantle_l, vp_l, vs_l, h_l, n_l = [], [], [], [], [], [], []
    for i in range(24):
        n = int(rs.randint(1, 12))
        vsn = rs.uniform(2.0, 5.0, n)
        zn = np.sort(rs.uniform(0, 60, n))
        model = np.concatenate((vsn, zn))
        vpvs = rs.uniform(1.5, 2.0)
        mantle = None if i % 2 == 0 else [4.3, 1.8]
        vp, vs, h = Models.Model.get_vp_vs_h(model.copy(), vpvs, mantle)
        row = np.full(2 * Lmax, np.nan); row[:n] = vsn; row[Lmax:Lmax + n] = zn
        vec.append(row); vpvs_l.append(vpvs); mantle_l.append(0 if mantle is None else 1)
        n_l.append(n)
        vp_l.append(np.asarray(vp, float)); vs_l.append(np.asarray(vs, float)); h_l.append(np.asarray(h, float))
    np.savez_compressed(os.path.join(HERE, "model_golden.npz"), nuclei=np.array(vec),
                        n=np.array(n_l, dtype=np.int32), vpvs=np.array(vpvs_l),
                        use_mantle=np.array(mantle_l, dtype=np.int32), mantle=np.array([4.3, 1.8]),
                        vp=pad(vp_l, Lmax), vs=pad(vs_l, Lmax), h=pad(h_l, Lmax))
    # ---- 5. a recorded SingleChain run (the caller of the hot path), two noise-law set-ups ---------------
    from multiprocessing import sharedctypes
    SingleChain = sys.modules["BayHunter.SingleChain"].SingleChain
    xsw = period_sets["p21"]
    xrf = time_axes["n201"]
    st3 = lambda n: np.loadtxt(os.path.join(REF, "tutorial", "observed", "st3_%s.dat" % n)).T
    nrs = np.random.RandomState(99)
    ysw = st3("rdispph")[1] + nrs.normal(0, 0.012, xsw.size)
    yrf = st3("prf")[1] + nrs.normal(0, 0.005, xrf.size)
    out = {"xsw": xsw, "ysw": ysw, "xrf": xrf, "yrf": yrf}
    setups = {"exp": dict(priors=dict(vpvs=(1.4, 2.1), layers=(1, 10), vs=(2, 5), z=(0, 60), rfnoise_corr=(0.35, 0.75),
                                      rfnoise_sigma=(1e-5, 0.05), swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.1)),
                          init=dict(nchains=1, iter_burnin=1500, iter_main=700, acceptance=(40, 80), thickmin=0.1, lvz=0.1,
                                    hvz=None, rcond=None, maxmodels=400), seeds=(11, 12)),
              "gauss": dict(priors=dict(vpvs=1.73, layers=(1, 8), vs=(2, 5), z=(0, 60), mohoest=(30, 8), rfnoise_corr=0.9,
                                        rfnoise_sigma=(1e-5, 0.05), swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.1)),
                            init=dict(nchains=1, iter_burnin=1200, iter_main=500, acceptance=(40, 80), thickmin=0.1, lvz=None,
                                      hvz=None, rcond=1e-5, maxmodels=50000), seeds=(21,)),
              # BASELINE configs[0]: the tutorial's set-up reduced to its Rayleigh phase target (observed errors
              # given -> scaled-error law), priors / proposal widths of tutorial/config.ini, 1 chain
              "tut": dict(priors=dict(vpvs=(1.4, 2.1), layers=(1, 20), vs=(2, 5), z=(0, 60), mohoest=None,
                                      swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.05)),
                          init=dict(nchains=1, iter_burnin=1400, iter_main=600, propdist=(0.015, 0.015, 0.015, 0.005, 0.005),
                                    acceptance=(40, 80), thickmin=0.1, lvz=None, hvz=None, rcond=1e-5, maxmodels=50000),
                          seeds=(31,), swd_only=True)}
    ysw_err = SynthObs.compute_expnoise(st3("rdispph")[1], corr=0.0, sigma=0.012)
    out["ysw_err"] = ysw_err
    for name, su in setups.items():
        for seed in su["seeds"]:
            if su.get("swd_only"):
                jt = Targets.JointTarget(targets=[Targets.RayleighDispersionPhase(xsw, ysw, yerr=ysw_err)])
            else:
                t1 = Targets.RayleighDispersionPhase(xsw, ysw)
                t2 = Targets.PReceiverFunction(xrf, yrf)
                t2.moddata.plugin.set_modelparams(gauss=1.0, p=6.4)
                jt = Targets.JointTarget(targets=[t1, t2])
            tmp = tempfile.mkdtemp(prefix="bhchain_")
            os.makedirs(os.path.join(tmp, "data"))
            init = dict(su["init"]); init["savepath"] = tmp; init["station"] = "gold"
            iters = init["iter_burnin"] + init["iter_main"]
            nmodels = int(iters * np.max(init["acceptance"]) / 100.)
            maxlayers = int(su["priors"]["layers"][1]) + 1
            mk = lambda n: sharedctypes.RawArray("f", n)
            ntg = len(jt.targets)
            shared = [mk(nmodels * maxlayers * 2), mk(nmodels * (ntg + 1)), mk(nmodels), mk(nmodels * 2 * ntg), mk(nmodels)]
            for a in shared:
                np.frombuffer(a, dtype=np.float32).fill(np.nan)
            chain = SingleChain(targets=jt, chainidx=0, initparams=init, modelpriors=su["priors"], sharedmodels=shared[0],
                                sharedmisfits=shared[1], sharedlikes=shared[2], sharednoise=shared[3], sharedvpvs=shared[4],
                                random_seed=seed)
            chain.run_chain()
            key = "%s_s%d_" % (name, seed)
            out[key + "models"] = np.array(chain.chainmodels); out[key + "likes"] = np.array(chain.chainlikes)
            out[key + "misfits"] = np.array(chain.chainmisfits); out[key + "noise"] = np.array(chain.chainnoise)
            out[key + "vpvs"] = np.array(chain.chainvpvs); out[key + "iters"] = np.array(chain.chainiter)
            out[key + "propdist"] = np.array(chain.propdist); out[key + "accepted"] = np.array(chain.accepted)
            out[key + "proposed"] = np.array(chain.proposed)
            for ph in ("p1", "p2"):   # the files the reference wrote (SingleChain.py:665-690)
                for nm in ("models", "likes", "misfits", "noise", "vpvs"):
                    f = os.path.join(tmp, "data", "c000_%s%s.npy" % (ph, nm))
                    if os.path.exists(f):
                        out[key + "file_" + ph + nm] = np.load(f)
            print("chain", name, seed, "accepted rows", chain.chainlikes.size, "final logL", chain.chainlikes[-1],
                  "propdist", chain.propdist)
            shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(HERE, "chain_golden.npz"), **out)

    for f in sorted(os.listdir(HERE)):
        p = os.path.join(HERE, f)
        if os.path.isfile(p):
            print("%-22s %8d B" % (f, os.path.getsize(p)))


def merge_fixture():
    """tests/golden/merge_golden.npz: the reference's PlotFromStorage.save_final_distribution (src/Plotting.py:157-262)
    on a folder of three chains' main-phase files -- the two recorded `exp` chains of chain_golden.npz and a copy of
    the second one with halved likelihoods (the outlier at dev = 0.3) -- with maxmodels small enough that every
    chain is sub-sampled (the module-level RandomState(333) of Plotting.py).  Inputs are in chain_golden.npz;
    outputs: outlier list, outliers.dat text and the merged c_*.npy arrays."""
    Targets, Models, SynthObs = import_reference()
    utils = sys.modules["BayHunter.utils"]
    Plotting = sys.modules["BayHunter.Plotting"]
    g = np.load(os.path.join(HERE, "chain_golden.npz"))
    tmp = tempfile.mkdtemp(prefix="bhmerge_")
    data = os.path.join(tmp, "data")
    os.makedirs(data)
    for c, (key, scale) in enumerate((("exp_s11_", 1.0), ("exp_s12_", 1.0), ("exp_s12_", 0.5))):
        for ph in ("p1", "p2"):
            for nm in ("models", "likes", "misfits", "noise", "vpvs"):
                a = g[key + "file_" + ph + nm]
                if nm == "likes":
                    a = (a * scale).astype(a.dtype)
                np.save(os.path.join(data, "c%.3d_%s%s" % (c, ph, nm)), a)
    t1 = Targets.RayleighDispersionPhase(g["xsw"], g["ysw"])
    t2 = Targets.PReceiverFunction(g["xrf"], g["yrf"])
    jt = Targets.JointTarget(targets=[t1, t2])
    cfg = os.path.join(data, "gold_config.pkl")
    utils.save_config(jt, cfg, priors=dict(layers=(1, 10)), initparams=dict(station="gold"))
    obj = Plotting.PlotFromStorage(cfg)
    obj.save_final_distribution(maxmodels=300, dev=0.3)
    out = {"maxmodels": 300, "dev": 0.3, "outliers": np.asarray(obj.outliers, dtype=float),
           "outliers_dat": np.array(open(os.path.join(data, "outliers.dat")).read())}
    for nm in ("models", "likes", "misfits", "noise", "vpvs"):
        out["c_" + nm] = np.load(os.path.join(data, "c_%s.npy" % nm))
    np.savez_compressed(os.path.join(HERE, "merge_golden.npz"), **out)
    shutil.rmtree(tmp)
    print("merge_golden.npz: outliers", out["outliers"], "merged rows", out["c_likes"].size)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "merge":
        merge_fixture()
        sys.exit(0)
    main()
