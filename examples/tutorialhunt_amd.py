#!/usr/bin/env python3
"""The reference's tutorial (tutorial/tutorialhunt.py) on the MI355X engine: same targets, priors and
initparams dictionaries, same result files -- the forward models, the likelihood and the sampler's
iteration run on the GPU.

    python examples/tutorialhunt_amd.py [--chains 5] [--device-chains 0] [--out results]

--device-chains N > 0 runs N device-resident chains (Philox draws) instead of the reference-order chains.
Observed data: the st3 synthetic test data of the reference's tutorial (kept as fixtures in tests/golden/st3).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayhunter_amd as bh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=5)
ap.add_argument("--device-chains", type=int, default=0)
ap.add_argument("--burnin", type=int, default=2048 * 4)
ap.add_argument("--main", type=int, default=2048 * 2)
ap.add_argument("--out", default="results")
args = ap.parse_args()

obs = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "st3")
xsw, _ysw = np.loadtxt(os.path.join(obs, "st3_rdispph.dat")).T
xrf, _yrf = np.loadtxt(os.path.join(obs, "st3_prf.dat")).T
rs = np.random.RandomState(333)
ysw = _ysw + rs.normal(0, 0.012, xsw.size)          # tutorial: uncorrelated noise on the dispersion curve
yrf = _yrf + rs.normal(0, 0.005, xrf.size)

target1 = bh.RayleighDispersionPhase(xsw, ysw)
target2 = bh.PReceiverFunction(xrf, yrf)
target2.moddata.plugin.set_modelparams(gauss=1.0, water=0.01, p=6.4)
targets = bh.JointTarget(targets=[target1, target2])

priors = dict(vpvs=(1.4, 2.1), layers=(1, 20), vs=(2, 5), z=(0, 60), mohoest=(38, 4), rfnoise_corr=0.98,
              swdnoise_corr=0., rfnoise_sigma=(1e-5, 0.05), swdnoise_sigma=(1e-5, 0.05))
initparams = dict(nchains=args.chains, iter_burnin=args.burnin, iter_main=args.main, propdist=(0.015, 0.015, 0.015, 0.005, 0.005),
                  acceptance=(40, 45), thickmin=0.1, lvz=None, hvz=None, rcond=1e-5, station="st3", savepath=args.out,
                  maxmodels=5000)

if args.device_chains > 0:
    chains = bh.DeviceChains(targets, args.device_chains, initparams, priors, seed=1).run()
    path = chains.save()
    like = chains.state_host()["like"]
else:
    opt = bh.MCMC_Optimizer(targets, initparams=initparams, priors=priors, random_seed=None)
    path = opt.mp_inversion()
    like = np.array([c.currentlikelihood for c in opt.batch.chains])
# what the reference's tutorial does next with PlotFromStorage: outlier chains + merged posterior files c_*.npy
outliers = bh.save_final_distribution(path, maxmodels=100000, dev=0.05)   # `path` = <savepath>/data
print("result files in", path, "- outlier chains:", [int(o) for o in outliers])
print("final log-likelihood of the chains: median %.1f, best %.1f" % (np.median(like), like.max()))
