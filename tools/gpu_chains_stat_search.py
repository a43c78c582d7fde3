#!/usr/bin/env python3
"""Does the short root refinement (bh_engine_set_swd_search) change what the chains sample?  The same problem
sampled by N device-resident chains with the reference sequence and by N others (another seed) with the short one;
posterior summaries compared in units of their Monte-Carlo standard error, as tools/gpu_chains_stat.py does for the
device step against the reference-order chains.  Dev tool; prints a table.
    python tools/gpu_chains_stat_search.py [N=256] [burn=4000] [main=2000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayhunter_amd as bh
from bayhunter_amd.device_chains import DeviceChains

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
burn, main = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4000, 2000)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "chain_golden.npz"))
priors = dict(vpvs=(1.4, 2.1), layers=(1, 10), vs=(2, 5), z=(0, 60), rfnoise_corr=(0.35, 0.75), rfnoise_sigma=(1e-5, 0.05),
              swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.1))
init = dict(nchains=1, iter_burnin=burn, iter_main=main, acceptance=(40, 45), thickmin=0.1, lvz=0.1, hvz=None, rcond=None,
            maxmodels=main // 20)


def targets():
    t1 = bh.RayleighDispersionPhase(g["xsw"], g["ysw"])
    t2 = bh.PReceiverFunction(g["xrf"], g["yrf"])
    t2.moddata.plugin.set_modelparams(gauss=1.0, p=6.4)
    return bh.JointTarget([t1, t2])


def summaries(models, likes, noise, vpvs):
    n = np.array([bh.Model.split_modelparams(m)[0] for m in models])
    d = np.array([2.0, 10.0, 25.0, 40.0, 55.0])
    v = np.zeros((models.shape[0], d.size))
    for i, m in enumerate(models):
        _, vs, z = bh.Model.split_modelparams(m)
        v[i] = vs[np.argmin(np.abs(z[:, None] - d[None, :]), axis=0)]
    return np.concatenate(([likes.mean(), n.mean(), vpvs.mean(), noise[:, 1].mean(), noise[:, 2].mean(), noise[:, 3].mean()], v.mean(axis=0)))


names = ["logL", "nuclei", "vp/vs", "sigma_swd", "corr_rf", "sigma_rf", "vs(2km)", "vs(10km)", "vs(25km)", "vs(40km)", "vs(55km)"]
eng = bh.default_engine(0)
out = {}
RUNS = tuple((m, int(s_)) for m, s_ in (x.split(":") for x in os.environ.get("RUNS", "reference:77,fast:78,reference:79,fast:80,reference:81,fast:82").split(",")))
for search, seed in RUNS:
    eng.set_swd_search(search)
    t0 = time.time()
    dc = DeviceChains(targets(), N, init, priors, seed=seed, search=None).run()   # (the engine's setting, set above)
    s = dc.samples("p2")
    out[(search, seed)] = np.array([summaries(s["models"][:, c], s["likes"][:, c], s["noise"][:, c], s["vpvs"][:, c]) for c in range(N)])
    print("%-9s seed %d: %d chains, %d + %d iterations, %.0f s" % (search, seed, N, burn, main, time.time() - t0), flush=True)
eng.set_swd_search("reference")


def zscores(A, B):
    return np.array([(B[:, j].mean() - A[:, j].mean()) / np.sqrt(A[:, j].var(ddof=1) / N + B[:, j].var(ddof=1) / N) for j in range(len(names))])


# pooled: all reference runs against all short-refinement runs
R = np.concatenate([out[k] for k in RUNS if k[0] == "reference"]); F = np.concatenate([out[k] for k in RUNS if k[0] == "fast"])
nr, nf = R.shape[0], F.shape[0]
print("pooled: %d reference-sequence chains against %d short-refinement chains" % (nr, nf))
print("%-10s %12s %12s %10s" % ("quantity", "reference", "short", "diff/sem"))
zp = []
for j, nm in enumerate(names):
    z = (F[:, j].mean() - R[:, j].mean()) / np.sqrt(R[:, j].var(ddof=1) / nr + F[:, j].var(ddof=1) / nf)
    zp.append(z)
    print("%-10s %12.5g %12.5g %10.2f" % (nm, R[:, j].mean(), F[:, j].mean(), z))
print("largest |difference| = %.2f standard errors" % np.max(np.abs(zp)))
print("pairs of runs, largest |difference| in standard errors (the scatter between seeds of ONE mode is the yardstick):")
for i, a in enumerate(RUNS):
    for b in RUNS[i + 1:]:
        print("   %-9s %d  vs  %-9s %d : %.2f" % (a[0], a[1], b[0], b[1], np.max(np.abs(zscores(out[a], out[b])))))
