#!/usr/bin/env python3
"""Single-model latency of the compatibility path: SurfDisp.run_model / RFminiModRF.run_model /
JointTarget.evaluate through the host API, wall clock per call (median of N), next to the kernel time inside."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayhunter_amd as bh
from bayhunter_amd.synth import true_model, SWD_PERIODS, RF_TIME
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nlay, h, vp, vs, rho = true_model(10)
h, vp, vs, rho = h[:, 0], vp[:, 0], vs[:, 0], rho[:, 0]
eng = bh.default_engine(0)


def bench(name, fn):
    for _ in range(10):
        fn()
    eng.set_instrumentation(True, False)
    eng.timing_reset()
    t = []
    for _ in range(N):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    n, tot, fam = eng.timing_collect()
    eng.set_instrumentation(False, False)
    t = np.array(t) * 1e3
    print("%-46s wall median %.3f ms (min %.3f)   kernels %.3f ms  %s" % (name, np.median(t), t.min(), tot / max(n, 1),
          {k: round(v / max(n, 1), 3) for k, v in fam.items() if v}), flush=True)


for ref in ("rdispph", "ldispph", "rdispgr"):
    p = bh.SurfDisp(SWD_PERIODS, ref)
    bench("SurfDisp(%s, 30 periods).run_model, 10 layers" % ref, lambda: p.run_model(h, vp, vs, rho))
r = bh.RFminiModRF(RF_TIME, "prf")
r.set_modelparams(gauss=2.5, p=6.4)
bench("RFminiModRF(prf, 1024 samples).run_model", lambda: r.run_model(h, vp, vs, rho))
t1 = bh.RayleighDispersionPhase(SWD_PERIODS, p.run_model(h, vp, vs, rho)[1] * 0 + 3.5)
t2 = bh.PReceiverFunction(RF_TIME, r.run_model(h, vp, vs, rho)[1])
t2.moddata.plugin.set_modelparams(gauss=2.5, p=6.4)
t1.get_covariance = t1.valuation.get_covariance_nocorr
t2.get_covariance = t2.valuation.get_covariance_exp
jt = bh.JointTarget([t1, t2])
noise = np.array([0.0, 0.02, 0.5, 0.01])
bench("JointTarget([rdispph, prf]).evaluate", lambda: jt.evaluate(h=h, vp=vp, vs=vs, noise=noise))
