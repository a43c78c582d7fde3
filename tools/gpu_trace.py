#!/usr/bin/env python3
"""Per-wavefront timeline of the dispersion group kernel on the c2 batch (development tool):
how long every wavefront is resident, how many rounds it takes, and where it ran.
    python tools/gpu_trace.py [G] [J]       (0 = the planner's choice)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS
eng = E.Engine(0)
rs = np.random.RandomState(5)
yobs = 3.4 + 0.01 * SWD_PERIODS
which = os.environ.get("TARGETS", "RL")
spec = []
if "R" in which:
    spec.append(dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=2, igr=0))
if "L" in which:
    spec.append(dict(kind=E.TARGET_SWD, law=0, n=30, x=SWD_PERIODS, yobs=yobs, iwave=1, igr=0))
eng.set_targets(spec)
B = int(os.environ.get("B", "4096"))
nlay, h, vp, vs, rho = synth_models(rs, B, 10, lvz_frac=0.1)
noise = np.tile([0, 0.05] * len(spec), (B, 1))
G = int(sys.argv[1]) if len(sys.argv) > 1 else 0
J = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng.set_swd_group(G)
eng.set_swd_lookahead(J)
eng.set_instrumentation(True, False)
for _ in range(2):
    eng.evaluate_batch(nlay, h, vp, vs, noise)
eng.timing_reset()
for _ in range(3):
    eng.evaluate_batch(nlay, h, vp, vs, noise)
n, tot, fam = eng.timing_collect()
print("G %d J %d targets %s B %d: swd %.3f ms per call (untraced)" % (G, J, which, B, fam["swd"] / n))
eng.set_instrumentation(True, True)
eng.timing_reset()
eng.evaluate_batch(nlay, h, vp, vs, noise)
n, tot, fam = eng.timing_collect()
c = eng.debug_counters()
tr = eng.debug_trace()
t0 = tr[:, 0].min()
start = (tr[:, 0] - t0) / 100.0          # us
end = (tr[:, 1] - t0) / 100.0
dur = end - start
cyc = (tr[:, 2] & 0xffffffffff).astype(float)
gwid = (tr[:, 2] >> 40).astype(np.int64)          # wavefront index in the grid (blockIdx.x * wavefronts per workgroup + wave)
rounds = (tr[:, 3] & 0xffffffff).astype(int)
ifn = ((tr[:, 3] >> 32) & 0xf).astype(int)
hw = (tr[:, 3] >> 36).astype(np.int64)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = ((hw >> 13) & 7) + 8 * ((hw >> 16) & 15)
print("traced call: swd %.3f ms, waves %d, evals %d" % (fam["swd"], c[7], c[0]))
print("span of all waves: %.1f us; clock (cycles / wall): %.0f MHz" % (end.max(), np.median(cyc / dur)))
for name, k in (("Rayleigh", 2), ("Love", 1)):
    m = ifn == k
    if not m.any():
        continue
    print("%-8s waves %5d  start us min/med/max %.0f/%.0f/%.0f  dur us min/med/mean/max %.0f/%.0f/%.0f/%.0f  rounds min/med/max %d/%d/%d  kcycles/round med %.2f"
          % (name, m.sum(), start[m].min(), np.median(start[m]), start[m].max(), dur[m].min(), np.median(dur[m]), dur[m].mean(), dur[m].max(),
             rounds[m].min(), np.median(rounds[m]), rounds[m].max(), np.median(cyc[m] / rounds[m]) / 1e3))
print("resident wave-time / (kernel span x 1024 SIMDs): %.2f waves per SIMD on average" % (dur.sum() / (end.max() * 1024)))
# waves per SIMD id
key = ((se * 2 + sh) * 16 + cu) * 4 + simd
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc,se,sh,cu,simd) seen: %d; waves per SIMD histogram:" % u.size, np.bincount(cnt))
# per-round cost vs. co-residency: correlate wave duration with number of waves on the same SIMD
per = {k: v for k, v in zip(u, cnt)}
share = np.array([per[k] for k in key])
for s_ in sorted(set(share)):
    m = (share == s_) & (ifn == 2)
    if m.any():
        print("  Rayleigh waves on SIMDs with %d wave(s): n %d, kcycles/round med %.2f, dur med %.0f us" % (s_, m.sum(), np.median(cyc[m] / rounds[m]) / 1e3, np.median(dur[m])))
# partner type: for every Rayleigh wave, is there a Love wave on its SIMD?
love_simds = set(key[ifn == 1].tolist())
for nm, sel in (("with a Love wave", np.array([k in love_simds for k in key]) & (ifn == 2)), ("without", np.array([k not in love_simds for k in key]) & (ifn == 2))):
    if sel.any():
        print("  Rayleigh waves sharing a SIMD %s: n %d, dur us med/max %.0f/%.0f, kcycles/round med %.2f" % (nm, sel.sum(), np.median(dur[sel]), dur[sel].max(), np.median(cyc[sel] / rounds[sel]) / 1e3))
for name, o in (("R", 1), ("L", 4)):
    nw = max(1, (ifn == (2 if name == "R" else 1)).sum())
    print("  %s per-wave Mcycles: A %.2f  B %.2f  S %.2f" % (name, c[o] / nw / 1e6, c[o + 1] / nw / 1e6, c[o + 2] / nw / 1e6))

# the slowest wavefronts: what do they share a SIMD / a CU with?
cukey = key // 4
order = np.argsort(-dur)[:12]
print("slowest wavefronts: type rounds kcyc/round dur_us | partner on the SIMD (type rounds dur) | CU: n Rayleigh, n Love | xcc")
for i in order:
    mates = [j for j in np.flatnonzero(key == key[i]) if j != i]
    cu = np.flatnonzero(cukey == cukey[i])
    ms = " ".join("%s %d %.0f" % ("RL"[ifn[j] == 1], rounds[j], dur[j]) for j in mates) or "-"
    print("   %s %4d %6.2f %5.0f | %-14s | %d %d | %d" % ("RL"[ifn[i] == 1], rounds[i], cyc[i] / rounds[i] / 1e3, dur[i], ms,
          int((ifn[cu] == 2).sum()), int((ifn[cu] == 1).sum()), int((hw[i] >> 16) & 15)))
# per-CU composition vs speed of the Rayleigh waves
nr = np.array([int((ifn[cukey == c] == 2).sum()) for c in cukey])
for k in sorted(set(nr[ifn == 2])):
    m = (ifn == 2) & (nr == k)
    print("  Rayleigh waves on CUs with %d Rayleigh wave(s): n %4d, kcycles/round med %.2f max %.2f" % (k, m.sum(), np.median(cyc[m] / rounds[m]) / 1e3, (cyc[m] / rounds[m]).max() / 1e3))

# which grid wavefronts share a SIMD?  (placement map: does the dispatcher pair them regularly?)
if os.environ.get("MAP"):
    pairs = []
    for k in u:
        w = np.sort(gwid[key == k])
        pairs.append(tuple(w))
    pairs.sort()
    d = np.array([p[1] - p[0] for p in pairs if len(p) == 2])
    print("SIMD pairs (grid wavefront indices): first 24:", pairs[:24])
    print("index distance inside a pair: histogram of the 10 most common:", sorted(zip(*np.unique(d, return_counts=True)), key=lambda t: -t[1])[:10])
    single = sorted(p[0] for p in pairs if len(p) == 1)
    print("wavefronts alone on a SIMD:", single)
    np.save(os.environ["MAP"], np.column_stack((gwid, key, ifn, rounds, dur)))
