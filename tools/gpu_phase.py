import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS
eng = E.Engine(0)
rs = np.random.RandomState(5)
yobs = 3.4 + 0.01 * SWD_PERIODS
eng.set_targets([dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=2, igr=0),
                 dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=1, igr=0)])
B = int(os.environ.get("B", "4096"))
nlay, h, vp, vs, rho = synth_models(rs, B, 10, lvz_frac=0.1)
noise = np.tile([0, 0.05, 0, 0.05], (B, 1))
Js = [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else '1').split(',')]
for G, J in [(int(a), j) for a in (sys.argv[1] if len(sys.argv) > 1 else '9,10').split(',') for j in Js]:
    eng.set_swd_group(G)
    eng.set_swd_lookahead(J)
    eng.set_instrumentation(True, True)
    eng.evaluate_batch(nlay, h, vp, vs, noise)
    eng.timing_reset()
    eng.evaluate_batch(nlay, h, vp, vs, noise)
    n, tot, fam = eng.timing_collect()
    c = eng.debug_counters()
    nw = max(1, c[7] / 2)
    print('G', G, 'J', J, 'ms', round(fam['swd'], 3), 'evals', c[0], 'waves', c[7])
    for name, o in (('R', 1), ('L', 4)):
        a, b, s_ = c[o] / nw, c[o + 1] / nw, c[o + 2] / nw
        print('  %s per-wave Mcycles: A %.2f  B %.2f  S %.2f  total %.2f' % (name, a / 1e6, b / 1e6, s_ / 1e6, (a + b + s_) / 1e6))
