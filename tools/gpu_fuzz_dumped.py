#!/usr/bin/env python3
"""Dev tool: the models a fuzz run dumped (BH_FUZZ_DUMP=dir of tools/gpu_fuzz.py: bad_*.npz with the model, the call's shape, the
engine's row and the reference's) once more -- alone and in a batch of the dumped size, with every trial count of the
trial-per-lane kernel and with the short refinement in the reference's arithmetic: which settings reproduce the difference
(docs/HISTORY.md, round 6: how the models beyond tolerance were classified).
    python tools/gpu_fuzz_dumped.py DIR"""
import glob, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayhunter_amd import engine as E
eng = E.default_engine(0)
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bad_*.npz"))):
    d = np.load(f); n = int(d["nlay"]); B0 = int(d["B"])
    print(os.path.basename(f), "ref", d["ref"][:4], "k", int(d["k"]))
    for B in (1, B0):
        tile = lambda a: np.repeat(np.asarray(a, dtype=np.float64).reshape(-1, 1), B, axis=1)
        nlay = np.full(B, n, dtype=np.int32)
        for arith in ("fast", "exact"):
            eng.set_swd_search("fast"); eng.set_swd_arith(arith)
            for tr in ((0, 4, 8, 16, 32, 64) if arith == "fast" else (0,)):
                eng.set_swd_trials(tr)
                v, e = eng.swd_batch(nlay, tile(d["h"]), tile(d["vp"]), tile(d["vs"]), tile(d["rho"]), d["per"], int(d["iwave"]), 0, flsph=int(d["flsph"]))
                eng.set_swd_trials(0)
                rv = d["ref"]; both = (v[0] != 0) & (rv != 0)
                rel = np.zeros_like(rv); rel[both] = np.abs(v[0][both] - rv[both]) / np.abs(rv[both])
                print("  B", B, arith, "trials", tr, eng.last_swd_kernel(), "rel %.3g at k=%d" % (rel.max(), int(rel.argmax())), "guarded", sum(eng.guard_stats()[0]), "v[k..]", v[0][int(d["k"]) - 1:int(d["k"]) + 2], flush=True)
