#!/usr/bin/env python3
"""What the trial-per-lane kernel's round structure rests on, measured on the bench's models with the CPU oracle (dev tool):
  * how many steps of 0.005 km/s the reference's scan takes from its start value c(k-1) - 1.5 dc to the sign change, per period
    (a window of W lanes riding along with the previous period's cluster reaches it in the fraction printed);
  * how far the inverse-quadratic point through the bracket's ends and the grid point before them is from the root
    (the cluster beside the window is two lanes at x -+ 2e-7 |x|).
    python tools/cpu_scan_steps.py [models]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
np.seterr(all='ignore')
from bayhunter_amd.synth import synth_models, SWD_PERIODS
from oracle import oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
L = O.lib()
_f = C.POINTER(C.c_float)
rs = np.random.RandomState(7)
nlay, h, vp, vs, rho = synth_models(rs, B, 10, lvz_frac=0.1)
dc = float(np.float32(0.005))
t = np.ascontiguousarray(SWD_PERIODS)
for iwave, name in ((2, "Rayleigh"), (1, "Love")):
    steps, dist = [], []
    for b in range(B):
        m = [np.ascontiguousarray(a[:, b], np.float32) for a in (h, vp, vs, rho)]
        cg = np.zeros(t.size)
        if O.surfdisp96(h[:, b], vp[:, b], vs[:, b], rho[:, b], 10, 0, iwave, 1, 0, t.size, t, cg):
            continue
        p = [a.ctypes.data_as(_f) for a in m]

        def f(c, om):
            return (L.bho_dltar4(om / c, om, p[0], p[1], p[2], p[3], 10, 1) if iwave == 2
                    else L.bho_dltar1(om / c, om, p[0], p[2], p[3], 10, 1))
        steps.append(np.ceil((cg[1:] - (cg[:-1] - 1.5 * dc)) / dc))
        if b >= 60:
            continue
        for k in range(1, t.size):
            om = 2 * np.pi / t[k]
            grid = (cg[k - 1] - 1.5 * dc) + dc * np.arange(0, 48)
            v = np.array([f(c, om) for c in grid[:int(steps[-1][k - 1]) + 2]])
            n = next((i for i in range(1, v.size) if (v[i] < 0) != (v[0] < 0)), None)
            if n is None or n < 2:
                continue
            lo, hi = grid[n - 1], grid[n]
            for _ in range(60):
                mid = 0.5 * (lo + hi)
                if (f(mid, om) < 0) == (v[n - 1] < 0): lo = mid
                else: hi = mid
            x, y = grid[n - 2:n + 1], v[n - 2:n + 1]
            xi = (x[0] * y[1] * y[2] / ((y[0] - y[1]) * (y[0] - y[2])) + x[1] * y[0] * y[2] / ((y[1] - y[0]) * (y[1] - y[2]))
                  + x[2] * y[0] * y[1] / ((y[2] - y[0]) * (y[2] - y[1])))
            if not np.isfinite(xi):  # two equal values: the secant point
                xi = x[2] - y[2] * (x[2] - x[1]) / (y[2] - y[1])
            dist.append(abs(xi - 0.5 * (lo + hi)) / lo)
    s, d = np.array(steps), np.array(dist)
    print("%-8s steps to the sign change: mean %.1f; reached by a window of  8 lanes %.2f, 14 lanes %.2f, 30 lanes %.2f of the periods"
          % (name, s.mean(), (s <= 7).mean(), (s <= 13).mean(), (s <= 29).mean()))
    print("         rounds per period with 16 lanes: window of 8 %.2f, of 14 %.2f" % tuple(
        np.where(s <= w - 1, 1, 1 + np.ceil((s - (w - 1)) / 16.0)).mean() for w in (8, 14)))
    print("         |estimate - root| / root: median %.1e, 99 %% %.1e, worst %.1e; within 2e-7: %.4f   (%d periods)"
          % (np.median(d), np.quantile(d, .99), d.max(), (d <= 2e-7).mean(), d.size))
