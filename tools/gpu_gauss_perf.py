#!/usr/bin/env python3
"""The Gauss-law contraction alone (gauss_quad_kernel*, csrc/gauss_kernel.hip): bh_loglike_batch on device pointers for
B residual rows of n samples; time of the likelihood family (contraction + likelihood kernel) by HIP events, TFLOP/s of the
2 B n^2 contraction against the FP64 matrix peak.   python tools/gpu_gauss_perf.py [B] [n] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayhunter_amd import engine as E
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
eng = E.Engine(0)
rs = np.random.RandomState(1)
idx = np.arange(n)
R = 0.92 ** ((idx[:, None] - idx[None, :]).astype(float) ** 2)
rinv = np.linalg.pinv(R, rcond=1e-6)
yobs = rs.normal(0, 0.1, n)
eng.set_targets([{"kind": E.TARGET_USER, "law": E.LAW_GAUSS, "n": n, "yobs": yobs, "rinv": rinv, "logdet_r": float(np.linalg.slogdet(R)[1])}])
dev = torch.device("cuda", 0)
ymod = torch.from_numpy(yobs + rs.normal(0, 0.01, (B, n))).to(dev)
noise = torch.from_numpy(np.column_stack((np.full(B, 0.92), rs.uniform(0.005, 0.05, B)))).to(dev)
logL = torch.zeros(B, dtype=torch.float64, device=dev); mis = torch.zeros((B, 2), dtype=torch.float64, device=dev); err = torch.zeros(B, dtype=torch.int32, device=dev)
L = eng._L


def call():
    eng._check(L.bh_loglike_batch(eng._h, E.DEVICE, None, B, ymod.data_ptr(), None, noise.data_ptr(), logL.data_ptr(), mis.data_ptr(), err.data_ptr()))


for _ in range(5):
    call()
eng.synchronize()
eng.set_instrumentation(True, False)
eng.timing_reset()
for _ in range(reps):
    call()
nc, tot, fam = eng.timing_collect()
ms = fam["like"] / nc
d = ymod.cpu().numpy() - yobs
phi = np.einsum("bi,ij,bj->b", d[:64], rinv, d[:64])
s = noise.cpu().numpy()[:64, 1]
ref = -0.5 * (n * np.log(2 * np.pi) + 2 * n * np.log(s) + float(np.linalg.slogdet(R)[1])) - 0.5 * phi / s ** 2
print("B %d n %d tile %s: contraction + likelihood %.4f ms  -> %.1f TFLOP/s FP64 = %.1f %% of 78.6;  max rel err %.2e"
      % (B, n, os.environ.get("BH_GAUSS_TILE", "auto"), ms, 2.0 * B * n * n / (ms * 1e-3) / 1e12, 2.0 * B * n * n / (ms * 1e-3) / 1e12 / 78.6 * 100,
         np.max(np.abs(logL.cpu().numpy()[:64] - ref) / np.abs(ref))), flush=True)
