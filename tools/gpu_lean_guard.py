#!/usr/bin/env python3
"""Which rules fire the guard of the trial-per-lane kernel, on LVZ-rich ragged batches and on bench.py's batches.  Dev tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models
eng = E.Engine(0)
names = ["-", "water", "small start", "small scan", "step probes", "bracket probes", "betmx in bracket (both sides / next to it)", "root at end"]
def run(tag, nlay, h, vp, vs, rho, per, iwave):
    eng.set_instrumentation(False, True)
    v, e = eng.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0)
    c = eng.debug_counters()
    rs_ = [0] + [(c[14 if i <= 4 else 15] >> (16 * ((i - 1) & 3))) & 0xffff for i in range(1, 8)]
    print("%-34s iwave %d: failed %5d guarded %5d  " % (tag, iwave, int((e != 0).sum()), sum(eng.guard_stats()[0])) + ", ".join("%s %d" % (names[i], rs_[i]) for i in range(1, 8) if rs_[i]), flush=True)
per = np.linspace(2, 60, 30)
spec, batches, noise, truth, nrs = bench.build_workload("c2", 4096, 10, seed=20260927)
for ib, b in enumerate(batches):
    for iw in (2, 1):
        run("bench batch %d" % ib, *b[:5], per, iw)
rs = np.random.RandomState(1)
for L in (4, 12):
    nlay, h, vp, vs, rho = synth_models(rs, 4000, L, lvz_frac=0.25, ragged=True)
    for iw in (2, 1):
        run("ragged LVZ-rich, up to %d layers" % L, nlay, h, vp, vs, rho, per, iw)
