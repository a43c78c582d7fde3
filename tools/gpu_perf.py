#!/usr/bin/env python3
"""Dispersion kernel timing sweep over lanes-per-model G and batch size (development tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS
eng = E.Engine(0)
eng.set_instrumentation(True, False)
rs = np.random.RandomState(5)
yobs = 3.4 + 0.01 * SWD_PERIODS
spec = [dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=2, igr=0),
        dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=1, igr=0)]
eng.set_targets(spec)
Bs = [int(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else [4096]
Gs = [int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 4, 8, 16]
Js = [int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1]
LVZ = float(os.environ.get("LVZ", "0"))
ref = {}
for B in Bs:
    nlay, h, vp, vs, rho = synth_models(rs, B, 10, lvz_frac=LVZ) if LVZ else synth_models(rs, B, 10)
    noise = np.tile([0, 0.05, 0, 0.05], (B, 1))
    for G, J in [(G, J) for G in Gs for J in Js]:
        if G * J > 64 or (G == 1 and J > 1):
            continue
        eng.set_swd_group(G)
        eng.set_swd_lookahead(J)
        out = None
        eng.evaluate_batch(nlay, h, vp, vs, noise)
        eng.timing_reset()
        for rep in range(3):
            out = eng.evaluate_batch(nlay, h, vp, vs, noise, want_ymod=True)
        n, tot, fam = eng.timing_collect()
        key = B
        same = True
        if key in ref:
            same = np.array_equal(ref[key][0], out[0]) and np.array_equal(ref[key][3], out[3])
        else:
            ref[key] = out
        print('B', B, 'G', G, 'J', J, 'swd ms', round(fam['swd'] / n, 3), 'evals/s', int(B / (fam['swd'] / n * 1e-3)), 'identical to first G:', same, flush=True)
