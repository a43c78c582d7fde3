#!/usr/bin/env python3
"""Rounds of the trial-per-lane kernel on each batch bench.py rotates through (c2): looks for wavefronts that take far more rounds
than the rest.  Dev tool.   python tools/gpu_lean_slow.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from bayhunter_amd import engine as E
eng = E.Engine(0)
spec, batches, noise, truth, nrs = bench.build_workload("c2", 4096, 10, seed=20260927)
bench.observed_data(eng, spec, truth, nrs)
eng.set_targets(spec)
for ib, b in enumerate(batches):
    nlay, h, vp, vs, rho = b[:5]
    eng.set_instrumentation(True, True)
    eng.evaluate_batch(nlay, h, vp, vs, noise, rho=rho)
    eng.timing_reset()
    r = eng.evaluate_batch(nlay, h, vp, vs, noise, rho=rho)
    n, tot, fam = eng.timing_collect()
    c = eng.debug_counters()
    print("batch %d: swd %.3f ms; Rayleigh rounds mean %.1f most %d; Love mean %.1f most %d; failed %d" % (ib, fam["swd"], c[1] / 1024.0, c[3], c[4] / 1024.0, c[6], int((r[2] != 0).sum())), flush=True)
    if os.environ.get("FIND") and max(c[3], c[6]) > 150:
        # bisect for the slow model: halves of the batch
        idx = np.arange(nlay.size)
        while idx.size > 1:
            half = idx[: idx.size // 2]
            eng.evaluate_batch(nlay[half], h[:, half], vp[:, half], vs[:, half], noise[half], rho=rho[:, half])
            cc = eng.debug_counters()
            idx = half if max(cc[3], cc[6]) > 150 else idx[idx.size // 2:]
        m = int(idx[0])
        print("   slow model", m, "layers", nlay[m]); print("   h", h[:, m]); print("   vs", vs[:, m]); print("   vp", vp[:, m]); print("   rho", rho[:, m])
