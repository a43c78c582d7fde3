#!/usr/bin/env python3
"""Which rules fire the guard of the trial-per-lane kernel in a sampler's windows (bench.py's c4 / c5 chains), and how many
windows hold a guarded model at all (those pay the re-run launch).  Dev tool.
    python tools/gpu_chain_guard.py [chains=8] [windows=300]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bayhunter_amd as bh
from bayhunter_amd import engine as E
from bayhunter_amd.device_chains import DeviceChains
from bayhunter_amd.synth import true_model, SWD_PERIODS, RF_TIME, SEED
C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 300
eng = E.Engine(0)
nlay, h, vp, vs, rho = true_model(10)
nrs = np.random.RandomState(SEED + 2)
ys = {}
for name, iwave in (("r", 2), ("l", 1)):
    y, err = eng.swd_batch(nlay, h, vp, vs, rho, SWD_PERIODS, iwave, 0)
    ys[name] = y[0] + nrs.normal(0, 0.012, SWD_PERIODS.size)
yrf = eng.rf_batch(nlay, h, vp, vs, rho, 6.4, 2.5, 2048, 20.0, 5.0, 0, RF_TIME.size)[0] + nrs.normal(0, 0.005, RF_TIME.size)
t3 = bh.PReceiverFunction(RF_TIME, yrf)
t3.moddata.plugin.set_modelparams(gauss=2.5, p=6.4)
jt = bh.JointTarget([bh.RayleighDispersionPhase(SWD_PERIODS, ys["r"]), bh.LoveDispersionPhase(SWD_PERIODS, ys["l"]), t3], engine=eng)
priors = dict(vpvs=(1.4, 2.1), layers=(1, 20), vs=(2, 5), z=(0, 60), rfnoise_corr=(0.35, 0.75), rfnoise_sigma=(1e-5, 0.05),
              swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.1))
init = dict(iter_burnin=100000, iter_main=100000, acceptance=(40, 45), thickmin=0.1, lvz=None, hvz=None, rcond=None, maxmodels=10)
dc = DeviceChains(jt, C, init, priors, seed=20260927, device=0)
for _ in range(30):
    dc.iterate()
eng.synchronize(); torch.cuda.synchronize()
eng.set_instrumentation(False, True)
names = ["-", "water", "small / hinge start", "small scan", "step probes", "bracket probes", "special cell", "root at end"]
tot0 = np.array(eng.guard_totals())
nonempty = 0; models = 0; prev = tot0.copy()
rs_ = [0] * 8
it0 = dc.iiter
for w in range(NW):
    l0 = dc.launches
    dc.iterate()
    eng.synchronize(); torch.cuda.synchronize()
    t = np.array(eng.guard_totals())
    nonempty += int((t - prev).sum() > 0)
    prev = t
    c = eng.debug_counters()       # (the counters of the window's call: every call starts them anew)
    for i in range(1, 8):
        rs_[i] += (c[14 if i <= 4 else 15] >> (16 * ((i - 1) & 3))) & 0xffff
d = prev - tot0
print("%d chains, %d windows (%d iterations each): windows with a guarded model %d (%.0f %%); guarded models per target %s" %
      (C, NW, (dc.iiter - it0) // max(1, NW), nonempty, 100.0 * nonempty / NW, d[:3].tolist()))
print("reasons: " + ", ".join("%s %d" % (names[i], rs_[i]) for i in range(1, 8) if rs_[i]))
