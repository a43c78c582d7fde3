import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
eng = E.Engine(0); eng.set_instrumentation(True, False)
rs = np.random.RandomState(1)
for n in (201, 1024):
    idx = np.arange(n); R = 0.92 ** ((idx[:, None] - idx[None, :]).astype(float) ** 2)
    rinv = np.linalg.pinv(R, rcond=1e-6); ld = np.linalg.slogdet(R)[1]
    yobs = rs.normal(0, 0.1, n)
    eng.set_targets([{"kind": E.TARGET_USER, "law": E.LAW_GAUSS, "n": n, "yobs": yobs, "rinv": rinv, "logdet_r": ld}])
    B = 4096
    ymod = yobs + rs.normal(0, 0.01, (B, n)); noise = np.column_stack((np.full(B, 0.92), rs.uniform(0.005, 0.05, B)))
    eng.loglike_batch(ymod, noise); eng.timing_reset()
    logL, misf, err = eng.loglike_batch(ymod, noise)
    nc, tot, fam = eng.timing_collect()
    d = ymod[:8] - yobs; ref = np.array([-0.5 * (n * np.log(2 * np.pi) + 2 * n * np.log(noise[b, 1]) + ld) - 0.5 * d[b] @ rinv @ d[b] / noise[b, 1] ** 2 for b in range(8)])
    print('n', n, 'like kernel ms', round(fam['like'], 3), 'GFLOP/s', round(2 * B * n * n / fam['like'] / 1e6, 1), 'max rel err', float(np.max(np.abs(logL[:8] - ref) / np.abs(ref))))
