#!/usr/bin/env python3
"""Statistical parity of the device-resident chain step with the reference-order chains: the same problem
sampled by N chains each way (different random streams), posterior summaries compared in units of their
Monte-Carlo standard error (between-chain scatter).  Dev tool; prints a table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayhunter_amd as bh
from bayhunter_amd.chains import ChainBatch
from bayhunter_amd.device_chains import DeviceChains

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
burn, main = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5000, 3000)
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


def vs_at(models, depths):
    """models [ns, 2*ML] reference rows -> vs at the given depths (Voronoi: nearest nucleus)."""
    out = np.zeros((models.shape[0], depths.size))
    for i, m in enumerate(models):
        n, vs, z = bh.Model.split_modelparams(m)
        out[i] = vs[np.argmin(np.abs(z[:, None] - depths[None, :]), axis=0)]
    return out


def summaries(models, likes, noise, vpvs):
    """per-chain means of a few posterior functionals; inputs [ns, ...] of ONE chain"""
    n = np.array([bh.Model.split_modelparams(m)[0] for m in models])
    d = np.array([2.0, 10.0, 25.0, 40.0, 55.0])
    v = vs_at(models, d).mean(axis=0)
    return np.concatenate(([likes.mean(), n.mean(), vpvs.mean(), noise[:, 1].mean(), noise[:, 2].mean(), noise[:, 3].mean()], v))


names = ["logL", "nuclei", "vp/vs", "sigma_swd", "corr_rf", "sigma_rf", "vs(2km)", "vs(10km)", "vs(25km)", "vs(40km)", "vs(55km)"]
t0 = time.time()
hb = ChainBatch(targets(), list(range(1000, 1000 + N)), init, priors).run()
path = hb.save("/tmp/stat_host")
H = []
for c in range(N):
    ld = lambda k: np.load(os.path.join(path, "c%.3d_p2%s.npy" % (c, k)))
    H.append(summaries(ld("models"), ld("likes"), ld("noise"), ld("vpvs")))
H = np.array(H)
t1 = time.time()
dc = DeviceChains(targets(), N, init, priors, seed=77).run()
s = dc.samples("p2")
D = np.array([summaries(s["models"][:, c], s["likes"][:, c], s["noise"][:, c], s["vpvs"][:, c]) for c in range(N)])
t2 = time.time()
print("reference-order chains %.0f s, device chains %.0f s, %d chains each, %d + %d iterations" % (t1 - t0, t2 - t1, N, burn, main))
print("%-10s %12s %12s %10s" % ("quantity", "host mean", "device mean", "diff/sem"))
worst = 0.0
for j, nm in enumerate(names):
    mh, md = H[:, j].mean(), D[:, j].mean()
    sem = np.sqrt(H[:, j].var(ddof=1) / N + D[:, j].var(ddof=1) / N)
    z = (md - mh) / sem
    worst = max(worst, abs(z))
    print("%-10s %12.5g %12.5g %10.2f" % (nm, mh, md, z))
print("largest |difference| = %.2f standard errors" % worst)
