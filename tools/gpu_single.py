import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS
eng = E.Engine(0)
eng.set_instrumentation(True, False)
rs = np.random.RandomState(5)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nlay, h, vp, vs, rho = synth_models(rs, B, 10)
for iwave, name in ((2, 'R'), (1, 'L')):
    for G in [int(a) for a in sys.argv[2].split(',')]:
        eng.set_swd_group(G)
        eng.swd_batch(nlay, h, vp, vs, rho, SWD_PERIODS, iwave, 0)
        eng.timing_reset()
        for rep in range(3):
            eng.swd_batch(nlay, h, vp, vs, rho, SWD_PERIODS, iwave, 0)
        n, tot, fam = eng.timing_collect()
        print(name, 'B', B, 'G', G, 'waves', (B + 64 // G - 1) // (64 // G), 'ms', round(fam['swd'] / n, 3), flush=True)
