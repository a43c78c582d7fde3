#!/usr/bin/env python3
"""Rounds and cycles per round of the trial-per-lane kernel (swd_lean.hip) on the c2 batch.  Dev tool.
    [BH_SWD_LEAN_R=16 BH_SWD_LEAN_L=8] python tools/gpu_lean_rounds.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = E.Engine(0)
rs = np.random.RandomState(5)
nlay, h, vp, vs, rho = synth_models(rs, B, 10, lvz_frac=0.1)
if os.environ.get("MODELS", "") == "prior":   # what a sampler proposes: 2..20 layers, unsorted vs in (2, 5), interfaces anywhere in (0, 60) km
    Lm = 20
    nlay = rs.randint(2, Lm + 1, B).astype(np.int32)
    h = np.zeros((Lm, B)); vp = np.zeros((Lm, B)); vs = np.zeros((Lm, B)); rho = np.zeros((Lm, B))
    for b in range(B):
        n = int(nlay[b])
        z = np.sort(rs.uniform(0, 60, n - 1)) if n > 1 else np.zeros(0)
        hh = np.diff(np.concatenate(([0.0], z)))
        v = rs.uniform(2, 5, n); k = rs.uniform(1.4, 2.1)
        h[:n - 1, b] = np.maximum(hh, 0.1); vs[:n, b] = v; vp[:n, b] = v * k; rho[:n, b] = 0.32 * v * k + 0.77
yobs = 3.4 + 0.01 * SWD_PERIODS
eng.set_targets([dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=2, igr=0),
                 dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=1, igr=0)])
noise = np.tile([0, 0.05, 0, 0.05], (B, 1))
for rep in range(2):
    eng.set_instrumentation(True, True)
    eng.evaluate_batch(nlay, h, vp, vs, noise)
    eng.timing_reset()
    eng.evaluate_batch(nlay, h, vp, vs, noise)
    n, tot, fam = eng.timing_collect()
    c = eng.debug_counters()
print("dispersion family %.3f ms; evaluations R %d (%.1f per model and period) L %d (%.1f)" % (fam["swd"], c[8], c[8] / B / 30.0, c[9], c[9] / B / 30.0))
nw = c[7]
for nm, o in (("Rayleigh", 1), ("Love", 4)):
    print("  %-8s rounds: sum %d, most %d; cycles per round %.0f, of which the evaluation %.0f" % (nm, c[o], c[o + 2], c[o + 1] / max(c[o], 1), c[12 if o == 1 else 13] / max(c[o], 1)))
print("  wavefronts", nw)
