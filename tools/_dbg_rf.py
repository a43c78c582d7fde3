import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models
from oracle import oracle as O
eng = E.Engine(0)
rs = np.random.RandomState(5)
nlay, h, vp, vs, rho = synth_models(rs, 6, 10)
for nsamp, fs, nk in ((64, 2.0, 32), (2048, 20.0, 1024)):
    y = eng.rf_batch(nlay, h, vp, vs, rho, 6.4, 2.5, nsamp, fs, 5.0, 0, nk)
    yo = O.rf_batch(nlay, h.T, vp.T, vs.T, rho.T, 6.4, 2.5, nsamp, fs, 5.0, 0, nk)
    print(nsamp, "nan rows", np.isnan(y).any(axis=1), "max err", np.nanmax(np.abs(y - yo)) / np.abs(yo).max())
    print(y[0, :6], yo[0, :6])
