import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from oracle import oracle as O
eng = E.Engine(0)
rs = np.random.RandomState(3)
for lo, hi in ((0, 1e-8), (0, 0.126), (0.1, 0.9), (0.8, 2.5), (2.4, 10), (0, 40), (1e3, 1e5)):
    x = rs.uniform(lo, hi, 400000) * rs.choice([-1, 1], 400000)
    for op, oop, nm in ((8, 0, 'sin'), (9, 1, 'cos')):
        g = eng.probe_math(op, x); h = O.libm_probe(oop, x)
        bad = np.flatnonzero(g.view(np.int64) != h.view(np.int64))
        print(lo, hi, nm, 'mismatches', bad.size, [(float(x[i]), g[i].hex(), h[i].hex()) for i in bad[:3]])
