#!/usr/bin/env python3
"""What the receiver function costs the dispersion kernel in the fused call (c3 against c2), from the kernel's own wavefront
trace in the steady state of back-to-back steps: engine clock, cycles per round, and the distribution of the wavefronts'
start and END times.  (rocprofv3 --pmc serialises the dispatches of all queues and cannot see the two kernels together.)
Round-4 finding: clock and cycles per round are the same, the MEDIAN wavefront ends at the same time -- the TAIL is longer:
RF workgroups become resident where dispersion workgroups have ended (LDS) and then share SIMDs with exactly the longest
wavefronts, the ones that set the kernel's time.  Dev tool.    python tools/gpu_c3_tail.py"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS
eng = E.Engine(0)
rs = np.random.RandomState(5)
B = 4096
nlay, h, vp, vs, rho = synth_models(rs, B, 10, lvz_frac=0.1)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d = [t(a) for a in (nlay, h, vp, vs, rho)]
yobs = 3.4 + 0.01 * SWD_PERIODS
swd = [dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=2, igr=0),
       dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=1, igr=0)]
rf = dict(kind=E.TARGET_RF, law=E.LAW_EXP, n=1024, yobs=np.zeros(1024), waveno=0, nsamp=2048, p=6.4, gauss=2.5, fsamp=20.0, tshift=5.0)
for name, spec in (("c2", swd), ("c3", swd + [rf]), ("c2", swd), ("c3", swd + [rf])):
    eng.set_targets(spec)
    nt = len(spec)
    noise = t(np.tile([0, 0.05, 0, 0.05] + ([0.5, 0.02] if nt == 3 else []), (B, 1)))
    logL = torch.zeros(B, dtype=torch.float64, device=dev); mis = torch.zeros((B, nt + 1), dtype=torch.float64, device=dev); err = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def step():
        eng.evaluate_batch_dev(B, 10, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), B, 1,
                               noise.data_ptr(), logL.data_ptr(), mis.data_ptr(), err.data_ptr(), stream=st)
    for _ in range(150): step()          # steady state of back-to-back steps (~0.5 s)
    eng.set_instrumentation(True, True)  # the traced call follows the others without a pause
    eng.timing_reset()
    step()
    torch.cuda.synchronize()
    n, tot, fam = eng.timing_collect()
    tr = eng.debug_trace()
    eng.set_instrumentation(False, False)
    dur = (tr[:, 1] - tr[:, 0]) / 100.0; cyc = (tr[:, 2] & 0xffffffffff).astype(float)
    rounds = (tr[:, 3] & 0xffffffff).astype(float)
    t0 = tr[:, 0].min(); start = (tr[:, 0] - t0) / 100.0; end = (tr[:, 1] - t0) / 100.0
    ifn = ((tr[:, 3] >> 32) & 0xf).astype(int)
    print("   wavefront start us: min %.0f med %.0f p99 %.0f max %.0f | end us: med %.0f p99 %.0f max %.0f | rounds R med %d L med %d | dur R med %.0f L med %.0f"
          % (start.min(), np.median(start), np.percentile(start, 99), start.max(), np.median(end), np.percentile(end, 99), end.max(),
             np.median(rounds[ifn == 2]), np.median(rounds[ifn == 1]), np.median(dur[ifn == 2]), np.median(dur[ifn == 1])))
    print("%s: traced dispersion kernel %.3f ms; engine clock during it (wave cycles / wall): median %.0f MHz; kcycles per round median %.2f"
          % (name, fam["swd"], np.median(cyc / dur), np.median(cyc / rounds) / 1e3))
