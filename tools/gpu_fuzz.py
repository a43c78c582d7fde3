#!/usr/bin/env python3
"""Randomised parity sweep of the dispersion kernels against the CPU oracle: random batch sizes, depths,
raggedness, periods, wave/velocity types, modes, earth flattening, lane mappings, look-ahead and depth
hints (dev tool; the fixed cases live in tests/).
    python tools/gpu_fuzz.py SEED NCONFIG          the reference sequence: bit-identical to the oracle
    LEAN=1 python tools/gpu_fuzz.py SEED NCONFIG   the engine's defaults (short refinement + fast arithmetic: the trial-per-lane kernel
                                                   where it applies): tolerance and failure flags against the reference sequence
    FAST=1 python tools/gpu_fuzz.py SEED NCONFIG   the short refinement with its guard (bh_engine_set_swd_search): bit-identical
                                                   to ITS CPU restatement (oracle search mode 2), and against the reference
                                                   sequence: failure flags and zero patterns (both must be the reference's),
                                                   worst relative difference; how many models the guard re-ran
    PRIOR=1 ...                                    models drawn from a sampler's prior (bayhunter_amd.synth.prior_models) instead of sorted velocities
    SCAN=steps|counted ...                         every scan step evaluated / the counted Love scan wherever a Love target is
                                                   (default: the engine's BH_SCAN_AUTO)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, prior_models
from oracle import oracle as O

eng = E.Engine(0)
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncfg = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
bad = 0
LEAN = os.environ.get("LEAN", "0") == "1"      # the engine's defaults: short refinement + fast arithmetic, the planner's own launch
FAST = os.environ.get("FAST", "0") == "1" or LEAN
PRIOR = os.environ.get("PRIOR", "0") == "1"
eng.set_swd_search("fast" if FAST else "reference")
eng.set_swd_arith("fast" if LEAN else "exact")   # (FAST = 1 alone: the reference's arithmetic, compared bit for bit with its restatement)
if os.environ.get("SCAN", "auto") in ("steps", "counted"):     # (default: the engine's BH_SCAN_AUTO)
    eng.set_swd_scan(os.environ["SCAN"])
nguard = 0
nbad_dumped = 0
worst, flagdiff, zerodiff, nmodels = 0.0, 0, 0, 0
t0 = time.time()
for it in range(ncfg):
    B = int(rs.choice([1, 2, 7, 33, 64, 65, 200, 700, 1500, 2600]))   # (2600: above the batch size below which the depth hint is ignored)
    L = int(rs.choice([2, 3, 5, 8, 10, 13, 17, 21, 30]))
    ragged = bool(rs.rand() < 0.6) and L > 2
    nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=float(rs.choice([0.0, 0.2, 0.5])), ragged=ragged)
    if L > 10:
        h[:-1] *= 10.0 / L
    if PRIOR and L >= 2:   # models as a sampler proposes them: velocities in any order, thin layers (PRIOR=1)
        nlay, h, vp, vs, rho = prior_models(rs, B, L, nmin=2 if ragged else L)
    K = int(rs.choice([1, 5, 21, 30, 60]))
    per = np.sort(rs.uniform(1.0, 80.0, K)) if rs.rand() < 0.5 else np.linspace(2, 60, K)
    iwave, igr = int(rs.choice([1, 2])), int(rs.choice([0, 1]))
    mode = int(rs.choice([1, 1, 1, 2, 3]))
    flsph = int(rs.rand() < 0.25)
    G = int(rs.choice([0, 0, 0, 1, 3, 5, 9, 12, 16, 21]))
    J = int(rs.choice([0, 0, 1, 2, 3, 4, 7]))
    hint = int(rs.choice([0, 0, 3, 6, 12]))
    if LEAN:
        G = J = 0
        trials_set = int(rs.choice([0, 0, 0, 4, 8, 16, 32, 64]))          # trials per round: by the call's shape, or pinned
        eng.set_swd_trials(trials_set)
    eng.set_swd_group(G); eng.set_swd_lookahead(J); eng.set_typical_layers(hint)
    with O.swd_search(2 if FAST else 0):
        ov, oe, _ = O.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr, mode=mode, flsph=flsph)
    if rs.rand() < 0.5:
        v, e = eng.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr, mode=mode, flsph=flsph)
    else:
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d = [t(a) for a in (nlay, h, vp, vs, rho, per)]
        dv = torch.zeros((B, K), dtype=torch.float64, device=dev); de = torch.zeros(B, dtype=torch.int32, device=dev)
        eng.swd_batch_dev(B, L, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), B, 1, K,
                          d[5].data_ptr(), iwave, igr, dv.data_ptr(), de.data_ptr(), stream=torch.cuda.current_stream().cuda_stream,
                          mode=mode, flsph=flsph)
        torch.cuda.synchronize()
        v, e = dv.cpu().numpy(), de.cpu().numpy()
    ok = LEAN or (np.array_equal(e, oe) and np.array_equal(v, ov))   # (LEAN: no bit-level restatement; tolerance and flags below)
    if not ok:
        bad += 1
        print("MISMATCH", dict(B=B, L=L, ragged=ragged, K=K, iwave=iwave, igr=igr, mode=mode, flsph=flsph, G=G, J=J, hint=hint),
              "err diff", int((e != oe).sum()), "vel diff", int((v != ov).sum()), flush=True)
    if FAST:   # against the reference sequence
        nguard += sum(eng.guard_stats()[0])
        rv, re_, _ = O.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr, mode=mode, flsph=flsph)
        both = (v != 0) & (rv != 0)
        if both.any():
            w = float(np.max(np.abs(v[both] - rv[both]) / np.abs(rv[both])))
            if w > 1e-5:
                print("TOLERANCE", w, dict(B=B, L=L, K=K, iwave=iwave, igr=igr, mode=mode, flsph=flsph), flush=True)
            worst = max(worst, w)
        # (dev aid: the worst models of the configurations that miss the tolerance or a flag, for a look on the CPU: BH_FUZZ_DUMP=dir)
        if os.environ.get("BH_FUZZ_DUMP") and nbad_dumped < 12:
            rel = np.zeros_like(v); rel[both] = np.abs(v[both] - rv[both]) / np.abs(rv[both])
            badm = np.where((rel.max(axis=1) > 1e-5) | (e != re_) | ((v == 0) != (rv == 0)).any(axis=1))[0]
            for b_ in badm[:2]:
                os.makedirs(os.environ["BH_FUZZ_DUMP"], exist_ok=True)
                np.savez(os.path.join(os.environ["BH_FUZZ_DUMP"], "bad_%d_%d.npz" % (it, b_)), nlay=nlay[b_], h=h[:, b_], vp=vp[:, b_], vs=vs[:, b_], rho=rho[:, b_],
                         per=per, iwave=iwave, flsph=flsph, lean=v[b_], ref=rv[b_], elean=e[b_], eref=re_[b_], k=int(np.argmax(rel[b_])),
                         B=B, Lmax=L, trials=(trials_set if LEAN else -1))   # (B, Lmax, trials: the call's shape decides the trials per round)
                nbad_dumped += 1
        flagdiff += int((e != re_).sum())
        zerodiff += int(((v == 0) != (rv == 0)).any(axis=1).sum())
        nmodels += B
if FAST:
    print("against the reference sequence: %d models, worst relative difference %.3g, failure flags differing %d, rows with a "
          "different zero pattern %d, models re-run by the guard %d" % (nmodels, worst, flagdiff, zerodiff, nguard))
    bad += flagdiff + zerodiff
print("%d configurations, %d mismatches, %.0f s" % (ncfg, bad, time.time() - t0))
sys.exit(1 if bad else 0)
