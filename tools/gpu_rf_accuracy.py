#!/usr/bin/env python3
"""Receiver function against the oracle for several filter widths / lengths / wave types (max error relative to the
peak, NaN rows), and the time of the RF kernels for the c3 and the tutorial filter; BH_RF_NO_CUT=1 computes every bin
(dev tool: the cut-off must not change a digit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models
from oracle import oracle as O
eng = E.Engine(0)
eng.set_instrumentation(True, False)
rs = np.random.RandomState(5)
for gauss, nsamp, nkeep, wave in ((2.5, 2048, 1024, 0), (1.0, 2048, 1024, 0), (1.0, 512, 201, 1), (5.0, 1024, 512, 0), (0.5, 512, 256, 0)):
    B = 96
    nlay, h, vp, vs, rho = synth_models(rs, B, 10, ragged=True, lvz_frac=0.3)
    y = eng.rf_batch(nlay, h, vp, vs, rho, 6.4, gauss, nsamp, 20.0, 5.0, wave, nkeep)
    yo = O.rf_batch(nlay, h.T, vp.T, vs.T, rho.T, 6.4, gauss, nsamp, 20.0, 5.0, wave, nkeep)
    peak = np.abs(yo).max(axis=1, keepdims=True)
    ok = np.isfinite(yo).all(axis=1)
    print("gauss %.1f nsamp %4d wave %d: max |rf - oracle| / peak = %.3g  (nan rows equal: %s)" % (gauss, nsamp, wave, (np.abs(y - yo) / peak)[ok].max(), np.array_equal(np.isnan(y).any(axis=1), ~ok)), flush=True)
nlay, h, vp, vs, rho = synth_models(rs, 4096, 10, ragged=False)
for gauss in (2.5, 1.0):
    eng.rf_batch(nlay, h, vp, vs, rho, 6.4, gauss, 2048, 20.0, 5.0, 0, 1024)
    eng.timing_reset()
    for _ in range(5):
        eng.rf_batch(nlay, h, vp, vs, rho, 6.4, gauss, 2048, 20.0, 5.0, 0, 1024)
    n, tot, fam = eng.timing_collect()
    print("B 4096 nsamp 2048 gauss %.1f: rf kernels %.3f ms" % (gauss, fam["rf"] / n), flush=True)
