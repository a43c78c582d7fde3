#!/usr/bin/env python3
"""Chain-iteration throughput: host-driven lock-step chains vs the device-resident chain step (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bayhunter_amd as bh
from bayhunter_amd.chains import ChainBatch
from bayhunter_amd.device_chains import DeviceChains

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "chain_golden.npz"))
priors = dict(vpvs=(1.4, 2.1), layers=(1, 20), vs=(2, 5), z=(0, 60), rfnoise_corr=(0.35, 0.75),
              rfnoise_sigma=(1e-5, 0.05), swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.1))
init = dict(nchains=1, iter_burnin=200, iter_main=100, acceptance=(40, 45), thickmin=0.1, lvz=0.1, hvz=None,
            rcond=None, maxmodels=50)


def targets():
    t1 = bh.RayleighDispersionPhase(g["xsw"], g["ysw"])
    t2 = bh.PReceiverFunction(g["xrf"], g["yrf"])
    t2.moddata.plugin.set_modelparams(gauss=1.0, p=6.4)
    return bh.JointTarget([t1, t2])


for C in (8, 64, 512):
    b = ChainBatch(targets(), list(range(C)), init, priors)
    t0 = time.perf_counter(); b.run(); dt = time.perf_counter() - t0
    print("host-driven ChainBatch  C=%5d: %7.2f ms/iteration  %9.0f chain-iterations/s" % (C, dt / 300 * 1e3, C * 300 / dt), flush=True)
for C in (8, 64, 512, 4096, 16384, 32768, 65536):
    d = DeviceChains(targets(), C, dict(init, maxmodels=5), priors, seed=1)   # a snapshot every 20 iterations
    d.engine.synchronize()
    t0 = time.perf_counter(); d.run(); dt = time.perf_counter() - t0
    st = d.state_host()
    print("device-resident chains  C=%5d: %7.2f ms/iteration  %9.0f chain-iterations/s   median logL %.1f" %
          (C, dt / 300 * 1e3, C * 300 / dt, np.median(st["like"])), flush=True)
