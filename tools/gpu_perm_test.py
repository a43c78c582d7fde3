#!/usr/bin/env python3
"""Does the ORDER of the models in a c2 batch change the dispersion kernel's time?  (development tool)
The kernel's time is that of its slowest SIMD (two wavefronts each); a wavefront's work is the longest root search among
its 3 (Rayleigh) / 7 (Love) models, and search length follows the model's S-velocity range (corr 0.94 / 0.80)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS
eng = E.Engine(0)
B, L = 4096, 10
rs = np.random.RandomState(20260927)
nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=0.1)
yobs = 3.4 + 0.01 * SWD_PERIODS
eng.set_targets([dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=2, igr=0),
                 dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=1, igr=0)])
eng.set_model_order(False)
dev = torch.device("cuda", 0)
cost = vs.max(axis=0) - vs.min(axis=0)
desc = np.argsort(-cost, kind="stable")


def blocks(order, g):            # order cut into blocks of g models
    return [order[i:i + g] for i in range(0, len(order), g)]


def alternate(order, g):         # block 0, last block, block 1, second-to-last, ...
    b = blocks(order, g); out = []
    i, j = 0, len(b) - 1
    while i <= j:
        out.append(b[i]); i += 1
        if i <= j:
            out.append(b[j]); j -= 1
    return np.concatenate(out)


def halves(order, g):            # first half descending, second half ASCENDING: block i pairs with block i + n/2 = its complement
    b = blocks(order, g); n = len(b) // 2
    return np.concatenate(b[:n] + b[n:][::-1])


perms = {"random (as generated)": np.arange(B), "descending cost": desc, "ascending cost": desc[::-1],
         "alternate blocks of 21": alternate(desc, 21), "alternate blocks of 84": alternate(desc, 84), "alternate blocks of 336": alternate(desc, 336),
         "halves, blocks of 21": halves(desc, 21), "shuffled blocks of 21 (homogeneous waves, random placement)": np.concatenate([blocks(desc, 21)[i] for i in rs.permutation(len(blocks(desc, 21)))])}
noise = torch.from_numpy(np.tile([0, 0.05, 0, 0.05], (B, 1))).to(dev)
logL = torch.zeros(B, dtype=torch.float64, device=dev); mis = torch.zeros((B, 3), dtype=torch.float64, device=dev); err = torch.zeros(B, dtype=torch.int32, device=dev)
ref = None
for name, p in perms.items():
    d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (nlay[p], h[:, p], vp[:, p], vs[:, p], rho[:, p])]
    def step():
        eng.evaluate_batch_dev(B, L, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), B, 1,
                               noise.data_ptr(), logL.data_ptr(), mis.data_ptr(), err.data_ptr())
    for _ in range(12):
        step()
    eng.synchronize()
    eng.set_instrumentation(True, False); eng.timing_reset()
    for _ in range(12):
        step()
    n, tot, fam = eng.timing_collect()
    eng.set_instrumentation(False, False)
    out = np.empty(B); out[p] = logL.cpu().numpy()
    if ref is None:
        ref = out
    print("%-62s swd %.4f ms   (same logL as the first order: %s)" % (name, fam["swd"] / n, np.array_equal(out, ref)), flush=True)
