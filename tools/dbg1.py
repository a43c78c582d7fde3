import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from oracle import oracle as O
np.set_printoptions(linewidth=200, precision=9)
eng = E.Engine(0); eng.set_instrumentation(False, True)
h = np.array([[2., 3., 10., 0.]]).T; vs = np.array([[3.9, 1.6, 4.4, 3.0]]).T; vp = vs * 1.75
rho = vp * 0.32 + 0.77
per = np.linspace(2, 60, 30)
for iwave, igr in ((2,0),(2,1),(1,0),(1,1)):
    v, e = eng.swd_batch(np.array([4]), h, vp, vs, rho, per, iwave, igr); ne = eng.last_neval()
    ov, oe, one = O.swd_batch(np.array([4]), h.T, vp.T, vs.T, rho.T, per, iwave, igr)
    print(iwave, igr, e, oe, ne, one)
    print(' gpu', v[0]); print(' cpu', ov[0]); print(' rel', np.abs(v[0]-ov[0])/np.maximum(np.abs(ov[0]),1e-30))
