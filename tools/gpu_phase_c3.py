#!/usr/bin/env python3
"""Where the receiver function's work beside the dispersion kernel goes (c3 against c2): the dispersion kernel's own phase
clocks -- wave-cycles spent in the layer terms (A), the recursion (B) and the search state machine (S) per wavefront and
round, Rayleigh and Love -- with and without the RF stream beside it.  rocprofv3 --pmc serialises the dispatches of all
queues, so hardware counters cannot see the two kernels together; the kernel's s_memtime clocks can.  Dev tool.
    python tools/gpu_phase_c3.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS, RF_TIME
eng = E.Engine(0)
rs = np.random.RandomState(5)
B = 4096
nlay, h, vp, vs, rho = synth_models(rs, B, 10, lvz_frac=0.1)
yobs = 3.4 + 0.01 * SWD_PERIODS
swd = [dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=2, igr=0),
       dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=1, igr=0)]
rf = dict(kind=E.TARGET_RF, law=E.LAW_EXP, n=RF_TIME.size, yobs=np.zeros(RF_TIME.size), waveno=0, nsamp=2048, p=6.4, gauss=2.5, fsamp=20.0, tshift=5.0)
for name, spec in (("c2", swd), ("c3", swd + [rf])):
    eng.set_targets(spec)
    noise = np.tile([0, 0.05, 0, 0.05] + ([0.5, 0.02] if len(spec) == 3 else []), (B, 1))
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
        res.append((fam["swd"], c, rounds, ifn))
    ms = np.median([r[0] for r in res])
    fam_swd, c, rounds, ifn = res[-1]
    print("%s: dispersion kernel %.3f ms (instrumented build, median of 3)" % (name, ms))
    for nm, o, k in (("Rayleigh", 1, 2), ("Love", 4, 1)):
        nr = rounds[ifn == k].sum()
        a, b, s_ = c[o] / nr, c[o + 1] / nr, c[o + 2] / nr
        print("   %-8s kcycles per wavefront-round: layer terms %.2f  recursion %.2f  state machine %.2f  sum %.2f   (wavefronts %d, rounds %d)"
              % (nm, a / 1e3, b / 1e3, s_ / 1e3, (a + b + s_) / 1e3, (ifn == k).sum(), nr))
