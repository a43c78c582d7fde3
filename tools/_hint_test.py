import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS
eng = E.Engine(0)
rs = np.random.RandomState(3)
B, L = 1016, 21
nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=0.1, ragged=True)
nlay = np.minimum(nlay, rs.randint(3, 10, B)).astype(np.int32)   # 3..9 layers, a few deeper
deep = rs.rand(B) < float(os.environ.get("DEEPFRAC", "0.05"))
nlay[deep] = rs.randint(10, 21, deep.sum())
for b in range(B):
    n = nlay[b]; h[n - 1, b] = 0.0; h[n:, b] = 0; vp[n:, b] = 0; vs[n:, b] = 0; rho[n:, b] = 0
yobs = 3.4 + 0.01 * SWD_PERIODS
eng.set_targets([dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=2, igr=0),
                 dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=1, igr=0)])
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d = [t(a) for a in (nlay, h, vp, vs, rho)]
noise = t(np.tile([0, 0.05, 0, 0.05], (B, 1)))
logL = torch.zeros(B, dtype=torch.float64, device=dev); mis = torch.zeros((B, 3), dtype=torch.float64, device=dev); err = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(hint, n=30):
    eng.set_typical_layers(hint)
    for _ in range(5):
        eng.evaluate_batch_dev(B, L, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), B, 1, noise.data_ptr(), logL.data_ptr(), mis.data_ptr(), err.data_ptr(), stream=st)
    torch.cuda.synchronize()
    eng.set_instrumentation(True, False); eng.timing_reset()
    for _ in range(n):
        eng.evaluate_batch_dev(B, L, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), B, 1, noise.data_ptr(), logL.data_ptr(), mis.data_ptr(), err.data_ptr(), stream=st)
    nc, tot, fam = eng.timing_collect()
    eng.set_instrumentation(False, False)
    return fam["swd"] / nc, float(logL.sum().item())
print("mean layers", nlay.mean(), "max", nlay.max())
for hint in (0, 6, 7, 8, 10, 12):
    ms, chk = run(hint)
    print("hint %2d: swd %.3f ms  (checksum %.6f)" % (hint, ms, chk))
