#!/usr/bin/env python3
"""The dispersion kernel's phase clocks with and without the certified-sign scan (c2, 4096 ten-layer models, default search):
rounds per wavefront, wave-cycles per round in the layer terms / recursion / state machine, and the look-ahead's own
statistics (counters 12-15: grid points evaluated, landings, wavefront-rounds with a look-ahead, their wave-cycles).  Dev tool.
    python tools/gpu_phase_prescan.py [search]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS
eng = E.Engine(0)
eng.set_swd_search(sys.argv[1] if len(sys.argv) > 1 else "fast")
rs = np.random.RandomState(5)
B = 4096
nlay, h, vp, vs, rho = synth_models(rs, B, 10, lvz_frac=0.1)
yobs = 3.4 + 0.01 * SWD_PERIODS
swd = [dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=2, igr=0),
       dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=1, igr=0)]
eng.set_targets(swd)
noise = np.tile([0, 0.05, 0, 0.05], (B, 1))
for pre in (0, 1):
    eng.set_swd_prescan(pre)
    res = []
    for rep in range(3):
        eng.set_instrumentation(True, True)
        eng.evaluate_batch(nlay, h, vp, vs, noise)
        eng.timing_reset()
        eng.evaluate_batch(nlay, h, vp, vs, noise)
        n, tot, fam = eng.timing_collect()
        c = eng.debug_counters()
        tr = eng.debug_trace()
        rounds = (tr[:, 3] & 0xffffffff).astype(float); ifn = ((tr[:, 3] >> 32) & 0xf).astype(int)
        cyc = (tr[:, 2] & 0xffffffffff).astype(float)
        res.append((fam["swd"], c, rounds, ifn, cyc))
    ms = np.median([r[0] for r in res])
    fam_swd, c, rounds, ifn, cyc = res[-1]
    print("prescan %d: dispersion kernel %.3f ms (instrumented build, median of 3); evaluations R %d L %d" % (pre, ms, c[8], c[9]))
    for nm, o, k in (("Rayleigh", 1, 2), ("Love", 4, 1)):
        nr = rounds[ifn == k].sum()
        a, b, s_ = c[o] / nr, c[o + 1] / nr, c[o + 2] / nr
        print("   %-8s rounds/wavefront mean %.0f max %.0f; kcycles per wavefront: mean %.0f max %.0f; per round: layer terms %.2f  recursion %.2f  state machine %.2f  sum %.2f  (wavefronts %d)"
              % (nm, rounds[ifn == k].mean(), rounds[ifn == k].max(), cyc[ifn == k].mean() / 1e3, cyc[ifn == k].max() / 1e3, a / 1e3, b / 1e3, s_ / 1e3, (a + b + s_) / 1e3, (ifn == k).sum()))
    if pre:
        print("   look-ahead: %d grid points by pending models' lanes, %d landings, %d wavefront-rounds with a look-ahead, %.2f kcycles each (%.1f %% of all rounds)"
              % (c[12], c[13], c[14], c[15] / max(1, c[14]) / 1e3, 100.0 * c[14] / rounds.sum()))
