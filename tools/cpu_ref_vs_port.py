#!/usr/bin/env python3
"""Single-thread time per call of the COMPILED REFERENCE (oracle/_ref: surfdisp96.f via amdflang -O2,
rfmini via g++ -O2) against the CPU restatement (oracle/) on the same c2 / c3 inputs (SURVEY.md 8(d),
CPU baseline item 1).  Build container only (needs /root/reference-built oracle/_ref)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O, refshim as R
from bayhunter_amd.synth import synth_models, SWD_PERIODS, RF_TIME

rs = np.random.RandomState(20260927)
B, L = 200, 10
nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=0.1)
ht, vpt, vst, rhot = h.T.copy(), vp.T.copy(), vs.T.copy(), rho.T.copy()
per = SWD_PERIODS
for name, iwave in (("Rayleigh phase", 2), ("Love phase", 1)):
    t0 = time.perf_counter()
    for b in range(B):
        thk = np.zeros(100, np.float32); a = np.zeros(100, np.float32); bb = np.zeros(100, np.float32); r = np.zeros(100, np.float32)
        thk[:L], a[:L], bb[:L], r[:L] = ht[b], vpt[b], vst[b], rhot[b]
        t = np.zeros(60); t[:per.size] = per
        cg = np.zeros(60)
        R.surfdisp96(thk, a, bb, r, L, 0, iwave, 1, 0, per.size, t, cg)
    t_ref = (time.perf_counter() - t0) / B
    t0 = time.perf_counter()
    O.swd_batch(nlay, ht, vpt, vst, rhot, per, iwave, 0, nthreads=1)
    t_port = (time.perf_counter() - t0) / B
    print("%-15s reference %.1f us/model   restatement %.1f us/model   ratio %.2f" % (name, t_ref * 1e6, t_port * 1e6, t_port / t_ref))
n = RF_TIME.size
t0 = time.perf_counter()
for b in range(40):
    z = np.concatenate(([0.0], np.cumsum(ht[b][:-1])))
    kap = vpt[b][0] / vst[b][0]
    R.synrf(z, vpt[b].copy(), vst[b].copy(), rhot[b].copy(), np.full(L, 500.0), np.full(L, 225.0), 6.4, 2.5, 2048, 20.0, 5.0,
            vst[b][0], (2 - kap ** 2) / (2 - 2 * kap ** 2), "P")
t_ref = (time.perf_counter() - t0) / 40
t0 = time.perf_counter()
O.rf_batch(nlay[:40], ht[:40], vpt[:40], vst[:40], rhot[:40], 6.4, 2.5, 2048, 20.0, 5.0, 0, n, nthreads=1)
t_port = (time.perf_counter() - t0) / 40
print("%-15s reference %.2f ms/model   restatement %.2f ms/model   ratio %.2f" % ("P-RF nsamp 2048", t_ref * 1e3, t_port * 1e3, t_port / t_ref))
