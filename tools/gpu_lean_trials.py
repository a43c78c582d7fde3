#!/usr/bin/env python3
"""Step time of the c2 batch against the trial count of the trial-per-lane kernel, per batch size: what bh_swd_lean_trials'
thresholds were read from.  Dev tool.   python tools/gpu_lean_trials.py [B ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS
sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 32768, 65536]
eng = E.Engine(0)
eng.set_instrumentation(True, False)
yobs = 3.4 + 0.01 * SWD_PERIODS
eng.set_targets([dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=2, igr=0),
                 dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=1, igr=0)])
for B in sizes:
    rs = np.random.RandomState(5)
    nlay, h, vp, vs, rho = synth_models(rs, B, 10, lvz_frac=0.1)
    noise = np.tile([0, 0.05, 0, 0.05], (B, 1))
    row = []
    for J in (0, 4, 8, 16, 32, 64):
        if J * B * 2 > 64 * (1 << 16) * 4:
            row.append("   -  ")
            continue
        eng.set_swd_trials(J)
        for _ in range(2):
            eng.evaluate_batch(nlay, h, vp, vs, noise)
        eng.timing_reset()
        for _ in range(8):
            eng.evaluate_batch(nlay, h, vp, vs, noise)
        n, tot, fam = eng.timing_collect()
        row.append("%6.3f" % (fam["swd"] / 8))
    print("B %6d  dispersion ms/step: policy %s | J=4 %s  8 %s  16 %s  32 %s  64 %s" % (B, *row), flush=True)
