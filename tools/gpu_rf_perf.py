#!/usr/bin/env python3
"""Receiver-function kernels alone, ONE batch shape per run (so that a rocprofv3 pass over this command holds one
shape only: VERDICT r02 weak 4):  python tools/gpu_rf_perf.py [shape] [reps]
    c3      B 4096, 10 layers, nsamp 2048, a = 2.5, 20 Hz, 1024 kept   (BASELINE configs[2]; the default)
    tut     the same batch with the tutorial's filter a = 1
    t512u   B 16384, 10 layers, nsamp 512, a = 1, 5 Hz, 201 kept       (chain batches)
    t512r   B 16384, up to 21 layers ragged, nsamp 512
    n8192 / n16384   B 256, 10 layers, nsamp 8192 / 16384, a = 2.5, 20 Hz (long traces)
Device pointers (torch-owned buffers), HIP events of the engine's instrumentation around the kernel family."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models
SHAPES = {"c3": (4096, 10, False, 2048, 1024, 2.5, 20.0), "tut": (4096, 10, False, 2048, 1024, 1.0, 20.0),
          "t512u": (16384, 10, False, 512, 201, 1.0, 5.0), "t512r": (16384, 21, True, 512, 201, 1.0, 5.0),
          "n8192": (256, 10, False, 8192, 4096, 2.5, 20.0), "n16384": (256, 10, False, 16384, 8192, 2.5, 20.0)}
shape = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, L, ragged, nsamp, nkeep, gauss, fsamp = SHAPES[shape]
eng = E.Engine(0)
dev = torch.device("cuda", 0)
rs = np.random.RandomState(3)
nlay, h, vp, vs, rho = synth_models(rs, B, L, ragged=ragged)
d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (nlay, h, vp, vs, rho)]
out = torch.zeros((B, nkeep), dtype=torch.float64, device=dev)


def call():
    eng.rf_batch_dev(B, L, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), B, 1,
                     6.4, gauss, nsamp, fsamp, 5.0, 0, nkeep, out.data_ptr())


for _ in range(3):
    call()
eng.synchronize()
eng.set_instrumentation(True, False)
eng.timing_reset()
for _ in range(reps):
    call()
n, tot, fam = eng.timing_collect()
ms = fam["rf"] / n
print("shape %-6s B %6d Lmax %2d ragged %-5s nsamp %5d a %.1f waves %s: rf kernels %.4f ms per batch (%.3e RF/s)"
      % (shape, B, L, ragged, nsamp, gauss, os.environ.get("BH_RF_WAVES", "4"), ms, B / (ms * 1e-3)), flush=True)
