import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from bayhunter_amd.synth import synth_models, SWD_PERIODS
rs = np.random.RandomState(1)
B = 16384
nlay, h, vp, vs, rho = synth_models(rs, B, 10)
ht, vpt, vst, rhot = [np.ascontiguousarray(a.T) for a in (h, vp, vs, rho)]
yobs = 3.4 + 0.01 * SWD_PERIODS
T = [dict(kind=0, law=0, n=30, iwave=2, igr=0, x=SWD_PERIODS, yobs=yobs), dict(kind=0, law=0, n=30, iwave=1, igr=0, x=SWD_PERIODS, yobs=yobs)]
noise = np.tile([0, 0.05, 0, 0.05], (B, 1))
for nth in (1, 8, 32, 64, 128, 256):
    n = min(B, 512 * nth)
    O.joint_batch(nlay[:nth * 4], ht[:nth * 4], vpt[:nth * 4], vst[:nth * 4], rhot[:nth * 4], T, noise[:nth * 4], nthreads=nth)
    t = time.perf_counter(); O.joint_batch(nlay[:n], ht[:n], vpt[:n], vst[:n], rhot[:n], T, noise[:n], nthreads=nth); dt = time.perf_counter() - t
    print('threads', nth, 'models', n, 'evals/s', int(n / dt), 'per-thread', int(n / dt / nth))
