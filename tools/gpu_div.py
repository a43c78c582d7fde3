import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
eng = E.Engine(0)
rs = np.random.RandomState(1)
tot = 0; bad = 0
for rep in range(40):
    n = 1 << 20
    ea = rs.uniform(-400, 400, n); eb = rs.uniform(-400, 400, n)
    if rep % 4 == 1: ea = eb + rs.uniform(-60, 0.0, n)       # |a| <= |b| like the max-norm rescale
    if rep % 4 == 2: ea = np.clip(eb + rs.uniform(-3, 3, n), -400, 400)
    a = np.ldexp(rs.uniform(1, 2, n), ea.astype(int)) * rs.choice([-1, 1], n)
    b = np.ldexp(rs.uniform(1, 2, n), eb.astype(int)) * rs.choice([-1, 1], n)
    if rep % 4 == 3:  # awkward mantissas
        a = np.ldexp(1 + rs.randint(0, 8, n) * 2.0 ** -52, ea.astype(int)); b = np.ldexp(2 - rs.randint(1, 8, n) * 2.0 ** -52, eb.astype(int))
    pairs = np.column_stack((a, b)).ravel()
    f = eng.probe_math(6, pairs); t = eng.probe_math(7, pairs); ref = a / b
    tot += n; bad += int((f.view(np.int64) != t.view(np.int64)).sum())
    assert np.array_equal(t, ref), 'hardware division differs from IEEE?'
print('pairs', tot, 'fast != plain:', bad)
