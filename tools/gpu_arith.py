#!/usr/bin/env python3
"""The fast arithmetic of the short-refinement launches (bh_engine_set_swd_arith) against the exact one and against the
reference's sequence: velocities, failure flags, guarded models, kernel time.  Dev tool.
    python tools/gpu_arith.py [B] [nrep]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = E.Engine(0)
if os.environ.get("GROUP"): eng.set_swd_group(int(os.environ["GROUP"]))      # 1 = the lane-per-evaluation kernel
if os.environ.get("LOOK"): eng.set_swd_lookahead(int(os.environ["LOOK"]))    # trial lanes per model
for seed, lvz in ((5, 0.1), (11, 0.5)):
    rs = np.random.RandomState(seed)
    nlay, h, vp, vs, rho = synth_models(rs, B, 10, lvz_frac=lvz)
    yobs = 3.4 + 0.01 * SWD_PERIODS
    spec = [dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=2, igr=0),
            dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=1, igr=0)]
    eng.set_targets(spec)
    noise = np.tile([0, 0.05, 0, 0.05], (B, 1))
    out = {}
    for search, arith in (("reference", "exact"), ("fast", "exact"), ("fast", "fast")):
        eng.set_swd_search(search)
        eng.set_swd_arith(arith)
        eng.set_instrumentation(True, False)
        r = eng.evaluate_batch(nlay, h, vp, vs, noise, want_ymod=True)
        eng.timing_reset()
        for _ in range(nrep):
            r = eng.evaluate_batch(nlay, h, vp, vs, noise, want_ymod=True)
        n, tot, fam = eng.timing_collect()
        out[(search, arith)] = r
        print("seed %d lvz %.1f  %-9s %-5s  swd kernel %.3f ms   guarded %s" % (seed, lvz, search, arith, fam["swd"] / max(n, 1), eng.guard_stats()[0][:2]), flush=True)
    ref = out[("reference", "exact")]
    for key in (("fast", "exact"), ("fast", "fast")):
        r = out[key]
        y0, y1 = np.asarray(ref[3]), np.asarray(r[3])
        e0, e1 = np.asarray(ref[2]), np.asarray(r[2])
        ok = (e0 == 0) & (e1 == 0)
        rel = np.abs(y1[ok] - y0[ok]) / np.maximum(np.abs(y0[ok]), 1e-300)
        zeros_same = np.array_equal(y0 == 0, y1 == 0)
        print("   %s/%s vs reference: flags equal %s, zero pattern equal %s, max rel velocity %.3e" % (key[0], key[1], np.array_equal(e0, e1), zeros_same, rel.max() if rel.size else 0.0))
    a, b = np.asarray(out[("fast", "exact")][3]), np.asarray(out[("fast", "fast")][3])
    nz = a != 0
    print("   fast/fast vs fast/exact: max rel %.3e, rows differing %d" % (np.max(np.abs(a[nz] - b[nz]) / np.abs(a[nz])), int((a != b).any(axis=1).sum())))
