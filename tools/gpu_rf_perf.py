#!/usr/bin/env python3
"""Receiver-function kernels alone: time per batch for a uniform and a ragged (transdimensional) batch (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models
eng = E.Engine(0)
eng.set_instrumentation(True, False)
rs = np.random.RandomState(3)
for B, L, ragged, nsamp, nkeep in ((4096, 10, False, 2048, 1024), (16384, 21, True, 512, 201), (16384, 10, False, 512, 201)):
    nlay, h, vp, vs, rho = synth_models(rs, B, L, ragged=ragged)
    eng.rf_batch(nlay, h, vp, vs, rho, 6.4, 1.0, nsamp, 20.0, 5.0, 0, nkeep)
    eng.timing_reset()
    for _ in range(3):
        eng.rf_batch(nlay, h, vp, vs, rho, 6.4, 1.0, nsamp, 20.0, 5.0, 0, nkeep)
    n, tot, fam = eng.timing_collect()
    print("B %6d Lmax %2d ragged %-5s nsamp %4d: rf kernels %.3f ms per batch (%.2e RF/s)" % (B, L, ragged, nsamp, fam["rf"] / n, B / (fam["rf"] / n * 1e-3)), flush=True)
