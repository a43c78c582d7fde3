#!/usr/bin/env python3
"""bench.py -- forward-model + logL evaluations per second on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload all|c2|c3|c4|c5|c5_full|c2g|c3g] [--batch B]
                    [--search reference|fast|fast_rayleigh] [--full] [--out bench_full.json]

Rank 0 prints ONE COMPACT JSON line (a few KB; tests/test_host_logic.py holds it under 8 KB): the c2 headline -- the
configuration the metric is quoted on -- with `roofline`, `cpu_baseline`, a five-key `parity_check` and a `summary` of
[value, ms/step] for the other BASELINE configs timed the same way right after it (c3 = configs[2], c2g / c3g = the second
runs of SURVEY.md 8(d), c4 / c5 / c5_full = configs[3] / [4]: device-resident chains, chain-iterations/s).  Everything
else every block measured goes to the file named by `full_record` (default bench_full.json beside this script).

--full adds what the default run leaves out to stay under a minute: the same workloads again with the other dispersion
search ("reference_search"), the receiver-function kernels alone ("rf_roofline"), the oracle port as a second CPU baseline
and the sweep over CPU pool sizes.

The line is measured with the engine's DEFAULT dispersion search (--search fast, bh_engine.h: the reference's brackets, a
three-evaluation refinement inside them; velocities within 1.2e-6 relative of the reference's -- north_star's tolerance is
1e-5 --, failure flags the reference's); `parity_check` states the velocities' largest relative difference against the
oracle's restatement of the REFERENCE's sequence.  --search reference measures with the reference's own sequence
(bit-identical velocities).

A "step" is ONE pass of the hot path over one batch of B synthetic candidate models that are
already resident in HBM: `bh_evaluate_batch` (C ABI, memspace = device) = every registered
target's forward model + RMS misfits + log-likelihood for all B models.

Workloads (BASELINE.json configs):
  c2 (default, the configuration the metric is quoted on): joint Rayleigh + Love PHASE dispersion,
     10-layer models, 30 periods linspace(2, 60, 30) s, batch = 4096 models / step / GPU,
     uncorrelated noise law.
  c3: c2 + P receiver function (Gauss a = 2.5, 1024 kept samples @ 20 Hz -> nsamp 2048,
     p = 6.4 s/deg), exponential-correlated noise law on the RF.
  c4 / c5 (BASELINE configs[3] / [4], the CALLER of the hot path): device-resident transdimensional
     chains (up to 20 layers) on the joint SWD + P-RF targets of c3; a "step" is one lock-step iteration
     of all chains = one forward+logL evaluation per chain.  c4: 8 chains per GPU.  c5: 64 chains per GPU
     at one temperature of an 8-rung ladder per rank, temperature exchange every 100 iterations (the
     all-gather of 3 floats per chain is the only collective).  Not the headline metric.
  c2g / c3g: the "second runs" of SURVEY.md 8(d) -- c2 with GROUP velocities; c3 with the Gauss law
     (fixed r = 0.92, rcond = 1e-6: the dense quadratic form on the FP64 matrix cores).
N > 1: one process per GPU (torchrun), independent batches per rank, no data-path collective
(the path shards by model) -> "scaling": "weak"; barrier + max-over-ranks timing.

`roofline` is for the dominant kernel (the dispersion kernel -- with the engine's defaults `swd_lean_kernel`, Rayleigh + Love
wavefronts in one launch): achieved = algorithmic bytes per launch / its average launch duration measured with HIP events on the launch
stream during the timed region.  `cpu_baseline` is the reference's own Fortran / C++ (oracle/_ref, built from
/root/reference where it exists; kind "reference") driven by a process pool on this box's host cores on a bounded sample
of the same workload; without oracle/_ref the oracle (the bit-exact CPU restatement, OpenMP) is the baseline (kind "port").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6    # CDNA4 FP64 vector peak; this path uses no MFMA
# flop-equivalents per layer-propagator step: Rayleigh ~320, Love ~65
FLOP_PER_LPS = {2: 320.0, 1: 65.0}
NPOOL = 4                   # distinct batches rotated through the steps
CLOCK_WARMUP_MS = float(os.environ.get("BH_BENCH_CLOCK_WARMUP_MS", "150"))   # untimed launches of the step before the --warmup steps (clock ramp of an idle GPU)
RF_CUT_WA = 12.5132         # csrc/rf_kernel.hip: w/a beyond which the Gauss low-pass is below 1e-17
PRIOR_LAYERS = 21           # c2p: array capacity of models drawn from the chains' prior (layers = (1, 20) above the half-space)
RF_FLOP_PER_LAYER_STEP = 600.0   # flop-equivalents of one layer of the reflectivity recursion for one frequency


def build_workload(name, B, L, seed):
    """Returns (targets for the engine, targets for the oracle, batches, noise) -- host arrays."""
    from bayhunter_amd import engine as E
    from bayhunter_amd.synth import synth_models, prior_models, true_model, SWD_PERIODS, RF_TIME, SEED
    rs = np.random.RandomState(seed)
    if name == "c2p":   # c2's targets on models drawn from the chains' prior (what configs[3] / [4] evaluate): ragged 2 .. L layers,
        batches = [prior_models(rs, B, L, vs=(2.0, 5.0), z=(0.0, 60.0), vpvs=(1.4, 2.1), thickmin=0.1) for _ in range(NPOOL)]   # velocities in any order
    else:
        batches = [synth_models(rs, B, L, lvz_frac=0.1) for _ in range(NPOOL)]
    per = SWD_PERIODS
    igr = 1 if name == "c2g" else 0     # c2g: the "second run" of SURVEY.md 8(d) with group velocities
    spec = [dict(kind=E.TARGET_SWD, law=E.LAW_NOCORR, n=per.size, x=per, iwave=2, igr=igr, name="rdispgr" if igr else "rdispph"),
            dict(kind=E.TARGET_SWD, law=E.LAW_NOCORR, n=per.size, x=per, iwave=1, igr=igr, name="ldispgr" if igr else "ldispph")]
    if name in ("c3", "c3g"):
        spec.append(dict(kind=E.TARGET_RF, law=E.LAW_EXP, n=RF_TIME.size, waveno=0, nsamp=2048, p=6.4,
                         gauss=2.5, fsamp=20.0, tshift=5.0, name="prf"))
    if name == "c3g":                   # second run of 8(d): Gauss law with fixed r = 0.92, rcond = 1e-6
        n = RF_TIME.size
        idx = np.arange(n)
        R = 0.92 ** ((idx[:, None] - idx[None, :]) ** 2.0)          # Targets.py:150-160
        spec[-1].update(law=E.LAW_GAUSS, rinv=np.linalg.pinv(R, rcond=1e-6), logdet_r=float(np.linalg.slogdet(R)[1]))
    nt = len(spec)
    noise = np.zeros((B, 2 * nt))
    for t, s in enumerate(spec):
        if s["law"] == E.LAW_EXP:
            noise[:, 2 * t] = rs.uniform(0.35, 0.75, B)
            noise[:, 2 * t + 1] = rs.uniform(1e-3, 0.05, B)
        elif s["law"] == E.LAW_GAUSS:
            noise[:, 2 * t] = 0.92
            noise[:, 2 * t + 1] = rs.uniform(1e-3, 0.05, B)
        else:
            noise[:, 2 * t + 1] = rs.uniform(0.005, 0.05, B)
    return spec, batches, noise, true_model(10 if name == "c2p" else L), np.random.RandomState(SEED + 2)


def observed_data(eng, spec, truth, nrs):
    """y_obs = engine forward model of the fixed 'true' model + N(0, sigma^2)."""
    from bayhunter_amd import engine as E
    nlay, h, vp, vs, rho = truth
    for s in spec:
        if s["kind"] == E.TARGET_SWD:
            y, err = eng.swd_batch(nlay, h, vp, vs, rho, s["x"], s["iwave"], s["igr"])
            assert err[0] == 0
            s["yobs"] = y[0] + nrs.normal(0, 0.012, s["n"])
        else:
            y = eng.rf_batch(nlay, h, vp, vs, rho, s["p"], s["gauss"], s["nsamp"], s["fsamp"], s["tshift"],
                             s["waveno"], s["n"])
            s["yobs"] = y[0] + nrs.normal(0, 0.005, s["n"])


_REF_JOB = None   # (nlay, h, vp, vs, rho [model-major], spec, noise) of a reference-baseline worker


def _ref_init(job):
    global _REF_JOB
    _REF_JOB = job


def _ref_slice(bounds):
    """One worker process: the COMPILED REFERENCE (oracle/_ref: surfdisp96.f, rfmini) on models lo..hi of
    the job, forward models through the reference's own code, likelihood through the dense restatement
    of Targets.py (oracle/like_oracle.c).  The Fortran keeps SAVE state: one process per worker."""
    from oracle import refshim as R, oracle as O
    from bayhunter_amd import engine as E
    nlay, ht, vpt, vst, rhot, spec, noise = _REF_JOB
    lo, hi = bounds
    acc = 0.0
    for b in range(lo, hi):
        i = b % nlay.size
        L = int(nlay[i])
        for t, s in enumerate(spec):
            if s["kind"] == E.TARGET_SWD:
                thk = np.zeros(100, np.float32); a = np.zeros(100, np.float32); bb = np.zeros(100, np.float32); r = np.zeros(100, np.float32)
                thk[:L], a[:L], bb[:L], r[:L] = ht[i, :L], vpt[i, :L], vst[i, :L], rhot[i, :L]
                per = np.zeros(60); per[:s["n"]] = s["x"]
                cg = np.zeros(60)
                R.surfdisp96(thk, a, bb, r, L, 0, s["iwave"], 1, s["igr"], s["n"], per, cg)
                ymod = cg[:s["n"]]
            else:
                z = np.concatenate(([0.0], np.cumsum(ht[i, :L - 1])))
                kap = vpt[i, 0] / vst[i, 0]
                rf = R.synrf(z, vpt[i, :L].copy(), vst[i, :L].copy(), rhot[i, :L].copy(), np.full(L, 500.0), np.full(L, 225.0),
                             s["p"], s["gauss"], s["nsamp"], s["fsamp"], s["tshift"], vst[i, 0], (2 - kap ** 2) / (2 - 2 * kap ** 2), "P")
                ymod = np.asarray(rf[-1] if isinstance(rf, tuple) else rf)[:s["n"]]
            acc += float(O.loglike_dense(s["law"], ymod, s["yobs"], noise[i, 2 * t], noise[i, 2 * t + 1], yerr=s.get("yerr")))
    return acc


def host_cpu_limits():
    """What this process may actually use of the box's CPUs: the scheduler affinity and the cgroup CPU quota (cgroup v2
    cpu.max, v1 cfs_quota / cfs_period); os.cpu_count() reports the machine, not the allowance."""
    out = {"os_cpu_count": os.cpu_count()}
    try:
        out["sched_affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        out["sched_affinity"] = None
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q < 0 else q / per
        except Exception:
            pass
    out["cgroup_cpu_quota"] = quota
    return out


def allowed_cpus():
    """The CPUs this process may use: min(scheduler affinity, cgroup quota), at most 64."""
    lim = host_cpu_limits()
    n = lim.get("sched_affinity") or lim["os_cpu_count"] or 1
    if lim.get("cgroup_cpu_quota"):
        n = min(n, max(1, int(round(lim["cgroup_cpu_quota"]))))
    return int(max(1, min(n, 64)))


def cpu_baseline_reference(spec, batch, noise, workload, worker_counts=None):
    """The compiled reference on this box's host cores (process pool), bounded to ~15-20 s of CPU work.
    worker_counts: the pool sizes to probe (default: one pool of allowed_cpus() workers; --full sweeps)."""
    global _REF_JOB
    import multiprocessing as mp
    nlay, h, vp, vs, rho = batch
    _REF_JOB = (nlay, np.ascontiguousarray(h.T), np.ascontiguousarray(vp.T), np.ascontiguousarray(vs.T),
                np.ascontiguousarray(rho.T), spec, noise)
    _ref_slice((0, 1))
    t0 = time.perf_counter()
    _ref_slice((0, 8))
    per_model = (time.perf_counter() - t0) / 8
    ncpu = os.cpu_count() or 1
    # fresh interpreters (no fork of a process that holds a HIP context and OpenMP threads)
    ctx = mp.get_context("spawn")

    def rate(nproc, n, reps=1):
        per = max(1, n // nproc)
        bounds = [(k * per, (k + 1) * per) for k in range(nproc)]
        with ctx.Pool(nproc, initializer=_ref_init, initargs=(_REF_JOB,)) as pool:
            pool.map(_ref_slice, [(0, 1)] * nproc)            # workers up, libraries loaded
            best_dt = None
            for _ in range(reps):                             # the box is shared: keep the least disturbed pass
                t0 = time.perf_counter()
                pool.map(_ref_slice, bounds, chunksize=1)
                dt = time.perf_counter() - t0
                best_dt = dt if best_dt is None else min(best_dt, dt)
            return per * nproc / best_dt

    if worker_counts is None:     # default: ONE pool of as many workers as this process may use (affinity / cgroup quota)
        worker_counts = (allowed_cpus(),)
    cands = sorted({c for c in worker_counts if c <= ncpu}) or [ncpu]
    if len(cands) > 1:
        probe = {c: rate(c, max(2 * c, int(0.3 * c / per_model / 4))) for c in cands}
        best = max(probe, key=probe.get)
        n = int(max(4 * best, min(20.0 / per_model, 6.0 * probe[best])))
    else:
        best = cands[0]
        probe = {}
        n = int(max(4 * best, 15.0 / per_model))                      # ~15 s of CPU work
    value = rate(best, n, reps=2)
    probe.setdefault(best, value)
    return {"value": value, "unit": "evals/s", "cores": best, "kind": "reference", "host_cpus": host_cpu_limits(),
            "rate_by_workers": {str(c): probe[c] for c in sorted(probe)},
            "sample": "%d models of the %s batch: forward models by the reference's own surfdisp96.f / rfmini compiled with "
                      "amdflang / g++ -O2 (oracle/_ref), dense likelihood as in Targets.py; %d worker processes (best of %s; "
                      "os.cpu_count() = %d), best of 2 passes; 1-process rate %.1f evals/s" % (n, workload, best, sorted(probe), ncpu, 1.0 / per_model)}


def cpu_baseline(spec, batch, noise, workload):
    """The oracle on this box's host cores, bounded to roughly 10-30 s of CPU work."""
    from oracle import oracle as O
    nlay, h, vp, vs, rho = batch
    ht, vpt, vst, rhot = [np.ascontiguousarray(a.T) for a in (h, vp, vs, rho)]
    ncpu = os.cpu_count() or 1
    probe = 16
    t0 = time.perf_counter()
    O.joint_batch(nlay[:probe], ht[:probe], vpt[:probe], vst[:probe], rhot[:probe], spec, noise[:probe], nthreads=1)
    per_model = (time.perf_counter() - t0) / probe            # single-thread seconds per evaluation

    def rate(nth, n):
        idx = np.arange(n) % nlay.size                        # tile the batch up to the sample size
        args = [a[idx] for a in (nlay, ht, vpt, vst, rhot)]
        O.joint_batch(*[a[:nth] for a in args], spec, noise[idx][:nth], nthreads=nth)   # spin the team up
        t0 = time.perf_counter()
        O.joint_batch(*args, spec, noise[idx], nthreads=nth)
        return n / (time.perf_counter() - t0)

    # The container may be allowed far fewer CPUs than os.cpu_count() reports (cgroup quota): take the
    # thread count that actually gives the highest rate on a short probe, then time the bounded sample.
    cands = sorted({max(1, ncpu >> k) for k in range(0, 4)} | {16, 32})
    cands = [c for c in cands if c <= ncpu]
    probe_rates = {c: rate(c, max(64, int(0.4 * c / per_model / 8))) for c in cands}
    best = max(probe_rates, key=probe_rates.get)
    n = int(max(4 * best, min(20.0 / per_model, 8.0 * probe_rates[best])))   # ~20 s of CPU work, <= ~8 s wall
    value = max(rate(best, n) for _ in range(3))                              # shared box: least disturbed pass
    return {"value": value, "unit": "evals/s", "cores": best, "kind": "port", "host_cpus": host_cpu_limits(),
            "sample": "%d models of the %s batch, all targets + dense logL, OpenMP over models with %d threads "
                      "(best of %s; os.cpu_count() = %d); 1-thread rate %.1f evals/s"
                      % (n, workload, best, sorted(probe_rates), ncpu, 1.0 / per_model)}


def run_chains(args, eng, rank, world, dist, dev, workload, steps, warmup):
    """c4 / c5: device-resident chains (bayhunter_amd.device_chains) on the c3 targets.  Returns the result block on
    rank 0 (None elsewhere).  `steps` / `warmup` are ITERATIONS of every chain (main phase timed, burn-in untimed); one
    evaluation launch advances every chain by `spec_depth` iterations (speculative windows, chain_kernel.hip)."""
    import torch
    import bayhunter_amd as bh
    from bayhunter_amd.device_chains import DeviceChains
    from bayhunter_amd.synth import true_model, SWD_PERIODS, RF_TIME, SEED
    full = workload == "c5_full"          # configs[4] whole on ONE GPU: 64 ladders x 8 temperatures = 512 chains (N = 1 only)
    C = args.chains or (8 if workload == "c4" else (512 if full else 64))
    nlay, h, vp, vs, rho = true_model(args.layers)
    nrs = np.random.RandomState(SEED + 2)
    ys = {}
    for name, iwave in (("r", 2), ("l", 1)):
        y, err = eng.swd_batch(nlay, h, vp, vs, rho, SWD_PERIODS, iwave, 0)
        ys[name] = y[0] + nrs.normal(0, 0.012, SWD_PERIODS.size)
    yrf = eng.rf_batch(nlay, h, vp, vs, rho, 6.4, 2.5, 2048, 20.0, 5.0, 0, RF_TIME.size)[0] + nrs.normal(0, 0.005, RF_TIME.size)
    t3 = bh.PReceiverFunction(RF_TIME, yrf)
    t3.moddata.plugin.set_modelparams(gauss=2.5, p=6.4)
    jt = bh.JointTarget([bh.RayleighDispersionPhase(SWD_PERIODS, ys["r"]), bh.LoveDispersionPhase(SWD_PERIODS, ys["l"]), t3], engine=eng)
    priors = dict(vpvs=(1.4, 2.1), layers=(1, 20), vs=(2, 5), z=(0, 60), rfnoise_corr=(0.35, 0.75), rfnoise_sigma=(1e-5, 0.05),
                  swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.1))
    init = dict(iter_burnin=warmup, iter_main=steps, acceptance=(40, 45), thickmin=0.1, lvz=None, hvz=None, rcond=None,
                maxmodels=max(1, steps // 100))
    kw = {}
    if workload == "c5":   # one temperature per rank (geometric ladder 1..30 over 8 rungs), ladders = chains
        ladder = 1.0 / np.geomspace(1.0, 30.0, 8)
        kw = dict(betas=np.full(C, ladder[rank % 8]), ladder=np.arange(C) + C * (rank // 8), swap_every=100)
    elif full:             # all eight temperatures of every ladder on this GPU: the exchange sweeps are inside the timed region
        ladder = 1.0 / np.geomspace(1.0, 30.0, 8)
        kw = dict(betas=np.tile(ladder, C // 8), ladder=np.repeat(np.arange(C // 8), 8) + (C // 8) * rank, swap_every=100)
    # ONE job seed on every rank: the chains' streams follow their global index, the exchange decisions the job seed
    dc = DeviceChains(jt, C, init, priors, seed=20260927, device=dev.index, dist=dist if world > 1 else None,
                      spec_depth=args.spec_depth or None, search=None, **({"arith": args.arith} if args.arith else {}), **kw)   # (search: the engine's setting = --search; arith: DeviceChains' own default unless --arith)

    def fence():
        eng.synchronize(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    while dc.iiter < 0:            # burn-in = warm-up, untimed
        dc.iterate()
    fence()
    eng.set_instrumentation(timing=True, counting=False)
    eng.timing_reset()
    fence()
    launches0 = dc.launches
    iter0 = dc.iiter               # (a speculative window may have carried the burn-in past iteration 0: count what is timed)
    guard0 = np.array(eng.guard_totals()) if eng.swd_search() != "reference" else np.zeros(8, dtype=np.int64)
    t0 = time.perf_counter()
    while dc.iiter < dc.iter_phase2:
        dc.iterate()
    fence()
    elapsed = time.perf_counter() - t0
    timed_iters = dc.iiter - iter0
    guard1 = np.array(eng.guard_totals()) if eng.swd_search() != "reference" else np.zeros(8, dtype=np.int64)
    ncalls, tot_ms, fam_ms = eng.timing_collect()
    eng.set_instrumentation(timing=False, counting=False)
    eng.set_typical_layers(0)      # (the engine is shared with the evaluate workloads that follow)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    nswaps = int(dc.nswaps)        # (collective-free: the swap decisions are identical on every rank)
    if rank != 0:
        return None
    st = dc.state_host()
    launches = dc.launches - launches0
    return {"metric": "chain-iterations/s (device-resident transdimensional chains)",
            "value": world * C * timed_iters / elapsed, "unit": "chain-iterations/s", "n_gpus": world, "steps": steps,
            "timed_iterations_per_chain": int(timed_iters),
            "warmup": warmup, "ms_per_step": elapsed / max(1, timed_iters) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if os.environ.get("BH_BENCH_DRYRUN", "0") != "1" else "synthetic (DRY RUN: all ranks on one GPU, gloo)",
            "config": {"workload": {"c4": "BASELINE configs[3]: independent chains sharded per GPU, joint Rayleigh+Love+P-RF, up to 20 layers",
                                    "c5": "BASELINE configs[4]: parallel tempering, one temperature of an 8-rung ladder per rank, exchange "
                                          "every 100 iterations, joint Rayleigh+Love+P-RF, up to 20 layers",
                                    "c5_full": "BASELINE configs[4] whole on one GPU: 64 ladders x 8 temperatures = 512 chains, exchange sweep "
                                               "(DeviceExchange, on the engine's stream) every 100 iterations inside the timed region, joint "
                                               "Rayleigh+Love+P-RF, up to 20 layers"}[workload],
                       "chains_per_gpu": C, "mean_layers_at_end": float(st["n"].mean()), "accepted_swaps": nswaps,
                       "parallelism": "chains sharded per GPU; c5: one small all-gather per exchange"},
            "search": {"mode": eng.swd_search(), "scan": eng.swd_scan(),
                       "models_rerun_by_the_guard": int((guard1 - guard0).sum()),
                       "models_rerun_by_target": [int(x) for x in (guard1 - guard0)[:2]],
                       "dispersion_models_evaluated": int(launches * C * ((1 << dc.depth) - 1) * 2)},
            "speculation": {"depth": dc.depth, "evaluation_launches": launches, "iterations_per_launch": timed_iters / max(1, launches),
                            "models_evaluated_per_launch": C * ((1 << dc.depth) - 1), "ms_per_launch": elapsed / max(1, launches) * 1e3,
                            "note": "the proposals of both outcomes of the next `depth` accept/reject decisions are evaluated in one "
                                    "launch and the realised path replayed: bit-identical to depth 1 (tests/test_gpu_device_chains.py)"},
            "reference_iterations_per_s_per_core": 357.0,
            "kernel_ms_per_launch": {k: v / max(1, ncalls) for k, v in fam_ms.items()},
            "median_logL": float(np.median(st["like"]))}


def parity_check(spec, batch, noise, d_logL, d_misf, d_err, n=64):
    """The outputs of the LAST timed step against the oracle on `n` models spread over the batch (the oracle as the
    checker, after the timed region): failure flags equal; largest relative (and absolute) difference of logL and of the
    misfits END TO END against the oracle's REFERENCE sequence -- at rounding level with --search reference; with the short
    refinement it is the velocities' 1e-6 (the reference's own roots are known to 1e-6: nevill's stop test) carried through
    the likelihood, see synthetics_check for the quantities north_star puts a tolerance on."""
    from oracle import oracle as O
    from bayhunter_amd import engine as E
    nlay, h, vp, vs, rho = batch
    B = nlay.size
    idx = np.unique(np.linspace(0, B - 1, n).astype(int))
    ht, vpt, vst, rhot = [np.ascontiguousarray(a.T[idx]) for a in (h, vp, vs, rho)]
    logL = d_logL.cpu().numpy()[idx]
    misf = d_misf.cpu().numpy()[idx]
    err = d_err.cpu().numpy()[idx]
    if any(s_["law"] == E.LAW_GAUSS for s_ in spec):      # the Gauss law is not in the oracle's batched entry point
        ref_L = np.zeros(idx.size)
        ref_m = None
        for k in range(idx.size):
            for t, s_ in enumerate(spec):
                a = [nlay[idx[k]:idx[k] + 1], ht[k:k + 1], vpt[k:k + 1], vst[k:k + 1], rhot[k:k + 1]]
                if s_["kind"] == E.TARGET_SWD:
                    y = O.swd_batch(*a, s_["x"], s_["iwave"], s_["igr"])[0][0]
                else:
                    y = O.rf_batch(*a, s_["p"], s_["gauss"], s_["nsamp"], s_["fsamp"], s_["tshift"], s_["waveno"], s_["n"])[0]
                ref_L[k] += O.loglike_dense(s_["law"], y, s_["yobs"], noise[idx[k], 2 * t], noise[idx[k], 2 * t + 1],
                                            rinv=s_.get("rinv"), logdet_r=s_.get("logdet_r", 0.0))
    else:
        ref_L, ref_m = O.joint_batch(nlay[idx], ht, vpt, vst, rhot, spec, noise[idx], nthreads=0)
    ok = err == 0
    out = {"n": int(idx.size), "failure_flags_equal": bool(np.array_equal(ok, ref_L > -1e14)),
           "max_rel_logL": float(np.max(np.abs(logL[ok] - ref_L[ok]) / np.abs(ref_L[ok]))) if ok.any() else None,
           "max_abs_logL": float(np.max(np.abs(logL[ok] - ref_L[ok]))) if ok.any() else None}
    if ref_m is not None and ok.any():
        out["max_rel_misfit"] = float(np.max(np.abs(misf[ok] - ref_m[ok]) / np.abs(ref_m[ok])))
    return out


def synthetics_check(spec, batch, noise, d_ymod, d_logL, d_err, n=64):
    """The synthetics of `n` models of the last step against the oracle, in north_star's own quantities and tolerances:
    dispersion velocities against the restatement of the REFERENCE's sequence (1e-5 relative; 0.0 = bit-identical), receiver
    functions (1e-4 of the trace's peak) -- over the models neither side fails on -- and the likelihood as a function of the
    synthetics: the oracle's dense likelihood of the DEVICE's synthetics against the device's logL (1e-8 relative).  With the
    reference search logL also agrees end to end (parity_check's max_rel_logL); with the short refinement the end-to-end
    difference is the velocities' 1e-6 carried through the likelihood, which has no tolerance of its own in north_star."""
    from oracle import oracle as O
    from bayhunter_amd import engine as E
    nlay, h, vp, vs, rho = batch
    idx = np.unique(np.linspace(0, nlay.size - 1, n).astype(int))
    a = [np.ascontiguousarray(x.T[idx]) for x in (h, vp, vs, rho)]
    ymod = d_ymod.cpu().numpy()[idx]
    logL = d_logL.cpu().numpy()[idx]
    err = d_err.cpu().numpy()[idx]
    worst, worst_rf, off = 0.0, None, 0
    for s_ in spec:
        if s_["kind"] == E.TARGET_SWD:
            ov, oe, _ = O.swd_batch(nlay[idx], *a, s_["x"], s_["iwave"], s_["igr"])
            ok = (oe == 0) & (err == 0)
            if ok.any():
                v = ymod[ok, off:off + s_["n"]]
                worst = max(worst, float(np.max(np.abs(v - ov[ok]) / np.abs(ov[ok]))))
        elif s_["kind"] == E.TARGET_RF:
            rf = O.rf_batch(nlay[idx], *a, s_["p"], s_["gauss"], s_["nsamp"], s_["fsamp"], s_["tshift"], s_["waveno"], s_["n"])
            ok = err == 0
            if ok.any():
                worst_rf = max(worst_rf or 0.0, float(np.max(np.abs(ymod[ok, off:off + s_["n"]] - rf[ok]) / np.abs(rf[ok]).max(axis=1, keepdims=True))))
        off += s_["n"]
    worst_l = 0.0
    for k in np.flatnonzero(err == 0)[:32]:
        o, off = 0.0, 0
        for t, s_ in enumerate(spec):
            o += O.loglike_dense(s_["law"], ymod[k, off:off + s_["n"]], s_["yobs"], noise[idx[k], 2 * t], noise[idx[k], 2 * t + 1],
                                 yerr=s_.get("yerr"), rinv=s_.get("rinv"), logdet_r=s_.get("logdet_r", 0.0))
            off += s_["n"]
        worst_l = max(worst_l, abs(logL[k] - o) / abs(o))
    out = {"max_rel_velocity": worst, "velocity_tolerance": 1e-5, "velocity_within_tolerance": bool(worst <= 1e-5),
           "max_rel_logL_of_the_device_synthetics": worst_l, "logL_of_synthetics_tolerance": 1e-8}
    if worst_rf is not None:
        out.update({"max_rf_over_peak": worst_rf, "rf_tolerance": 1e-4})
    return out


def pmc_summary(workload, B):
    """PMC figures of the dominant kernel for this workload and batch.  Counters cannot be read from inside the
    timed run: they come from the separate rocprofv3 --pmc passes of the same command (tools/profile_round.sh),
    whose summary is committed as profiles/pmc_summary.json; {} when there is no entry for this workload / batch."""
    try:
        d = json.load(open(os.path.join(REPO, "profiles", "pmc_summary.json")))[workload]
        if int(d["batch"]) == int(B):
            return d
    except Exception:
        pass
    return {}


def pmc_stamp(pmc, kernel_ms_now):
    """Provenance of the pasted counter figures: the commit and kernel time the committed pass was taken at, and whether this
    run's kernel time has moved away from it by more than 3 % (then the counter-derived fields describe an older build)."""
    if not pmc:
        return {"pmc_stale": None}
    k = pmc.get("kernel_ms")
    stale = None if (k is None or not kernel_ms_now) else bool(abs(k - kernel_ms_now) / kernel_ms_now > 0.03)
    return {"pmc_taken_at": {"commit": pmc.get("commit"), "kernel_ms": k, "tag": pmc.get("tag")}, "pmc_stale": stale}


def rf_roofline(eng, spec, d_batch, B, L, dev, reps=20):
    """The receiver-function kernels ALONE on the workload's own configuration: `reps` bh_rf_batch
    calls on device pointers, HIP events around the kernel family (engine instrumentation, on the launch stream);
    flop model = computed bins x (L-1) layer steps x 600 flop-equivalents + 5 N log2 N for the inverse FFT; HBM bytes =
    the model arrays in, the per-model coefficient record written and read once, nkeep samples out."""
    import torch
    from bayhunter_amd import engine as E
    s = [t for t in spec if t["kind"] == E.TARGET_RF][0]
    nl, h, vp, vs, rho = d_batch
    out = torch.zeros((B, s["n"]), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def call():
        eng.rf_batch_dev(B, L, nl.data_ptr(), h.data_ptr(), vp.data_ptr(), vs.data_ptr(), rho.data_ptr(), B, 1,
                         s["p"], s["gauss"], s["nsamp"], s["fsamp"], s["tshift"], s["waveno"], s["n"], out.data_ptr(), stream=stream)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    eng.set_instrumentation(timing=True, counting=False)
    eng.timing_reset()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    ncalls, tot_ms, fam = eng.timing_collect()
    eng.set_instrumentation(timing=False, counting=False)
    ms = fam["rf"] / max(1, ncalls)
    nsamp, half = s["nsamp"], s["nsamp"] // 2
    dw = 2.0 * np.pi * s["fsamp"] / nsamp
    jc = np.floor(RF_CUT_WA * s["gauss"] / dw) + 1.0                 # csrc/rf_kernel.hip: bh_launch_rf
    bins = half + 1 if (os.environ.get("BH_RF_NO_CUT") or not jc < half) else int(jc)
    flop = B * (bins * (L - 1) * RF_FLOP_PER_LAYER_STEP + 5.0 * nsamp * np.log2(nsamp))
    rec = 24 + 40 * L + 2                                            # doubles per coefficient record (rf_kernel.hip: rec_doubles)
    nbytes = B * (4 * L * 8 + s["n"] * 8)                            # the model in, the kept samples out
    workspace = B * 2 * rec * 8                                      # implementation traffic: the coefficient record written + read once
    tf = flop / (ms * 1e-3) / 1e12
    gbs = nbytes / (ms * 1e-3) / 1e9
    pmc = pmc_summary("rf_c3", B)
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": pmc.get("hbm_bytes_per_launch"), "traffic_unit": "bytes per call (coefficient + synthesis kernel)",
            "traffic_source": pmc.get("source"), **pmc_stamp(pmc, ms),
            "kernel": "rf_coef_layers_kernel + rf_synth_kernel, alone (no dispersion kernel beside them)", "kernel_ms_per_launch": ms,
            "algorithmic_bytes_per_launch": nbytes,
            "algorithmic_bytes_note": "SURVEY.md 8(d): B x (4 L x 8 B in + nkeep x 8 B out); the coefficient record the two kernels pass "
                                      "through HBM (workspace_bytes_per_launch) is implementation traffic and shows up under `traffic`",
            "workspace_bytes_per_launch": workspace,
            "config": {"B": B, "layers": L, "nsamp": nsamp, "gauss": s["gauss"], "fsamp": s["fsamp"], "nkeep": s["n"],
                       "bins_total": half + 1, "bins_computed": bins},
            "binding": {"bound": "fp64_valu", "achieved": tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s (flop-equivalents)",
                        "frac": tf / FP64_VALU_PEAK_TF, "flop_equivalents_per_layer_step": RF_FLOP_PER_LAYER_STEP,
                        "valu_busy": pmc.get("valu_busy"),
                        "note": "computed bins only (the bins the Gauss low-pass puts below 1e-17 are not computed, rf_kernel.hip)"},
            "rf_per_s": B / (ms * 1e-3)}


def run_eval(args, eng, rank, world, dist, dev, workload, dryrun, with_cpu=True, rf_roof=True):
    """One evaluate workload (c2 / c3 / c2g / c3g): untimed clock warm-up, `--warmup` steps, then EXACTLY `--steps`
    steps between barrier + synchronize on both sides, max over ranks.  Returns the result block on rank 0."""
    import torch
    from bayhunter_amd import engine as E
    B, L = args.batch, (PRIOR_LAYERS if workload == "c2p" else args.layers)
    spec, batches, noise, truth, nrs = build_workload(workload, B, L, seed=20260927 + 1000 * rank)
    observed_data(eng, spec, truth, nrs)
    eng.set_targets(spec)
    eng._owner = None
    if os.environ.get("BH_BENCH_AS_GIVEN", "0") == "1":   # (experiment switch: no on-device ordering of the batch)
        eng.set_model_order(sort_by_depth=False)
    nt = len(spec)

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_batches = [tuple(to_dev(a) for a in b) for b in batches]        # (nlay, h, vp, vs, rho) in HBM
    d_noise = to_dev(noise)
    d_logL = torch.zeros(B, dtype=torch.float64, device=dev)
    d_misf = torch.zeros((B, nt + 1), dtype=torch.float64, device=dev)
    d_err = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        nl, h, vp, vs, rho = d_batches[i % NPOOL]
        eng.evaluate_batch_dev(B, L, nl.data_ptr(), h.data_ptr(), vp.data_ptr(), vs.data_ptr(), rho.data_ptr(),
                               B, 1, d_noise.data_ptr(), d_logL.data_ptr(), d_misf.data_ptr(), d_err.data_ptr(),
                               stream=stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Clock warm-up: an idle GPU ramps its clock over the first launches (4.09 -> 3.60 ms over six
    # steps in the round-2 trace), longer than --warmup: ~CLOCK_WARMUP_MS of the same step, untimed, before the
    # warm-up steps.  Not part of K or W.
    eng.set_instrumentation(timing=False, counting=False)
    t0 = time.perf_counter()
    n_clock = 0
    while (time.perf_counter() - t0) * 1e3 < CLOCK_WARMUP_MS or n_clock < 4:
        step(n_clock)
        n_clock += 1
        if n_clock % 4 == 0:
            torch.cuda.synchronize()
    fence()
    eng.set_instrumentation(timing=False, counting=True)
    for i in range(args.warmup):
        step(i)
    fence()
    counters = eng.debug_counters() if args.warmup > 0 else None   # evaluations / layer steps of one step (last warm-up)
    eng.set_instrumentation(timing=True, counting=False)
    eng.timing_reset()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    ncalls, tot_ms, fam_ms = eng.timing_collect()
    # per-step times from the engine's own events (start of a call -> start of the next); NOT from further events on the
    # launch stream: a marker between two steps changed how the next step's two lane-kernel launches pair up on the SIMDs
    # (8.1 -> 10.6 ms per step at B = 16 384; no effect on the one-launch group kernel of B = 4096)
    per_step = eng.timing_steps()
    eng.set_instrumentation(timing=False, counting=False)
    eng.set_model_order(sort_by_depth=True)
    n_failed = int((d_err != 0).sum().item())
    # models of the LAST step that the short refinement's guard sent back to the reference's sequence (the second launch of the step)
    n_guarded = int(sum(eng.guard_stats()[0])) if eng.swd_search() != "reference" else 0
    finite = bool(torch.isfinite(d_logL).all().item())
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        tall = torch.zeros(world, dtype=torch.float64, device=dev)
        tall[rank] = elapsed
        dist.all_reduce(tall, op=dist.ReduceOp.SUM)
        rank_ms = [float(v) / args.steps * 1e3 for v in tall.tolist()]
        elapsed = float(tall.max().item())
    if rank != 0:
        return None
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    # dominant kernel: the dispersion kernel (all dispersion targets of a step in one launch);
    # algorithmic bytes per launch = (4*L*8 in + K*8 + 4 out) per model and target
    K = spec[0]["n"]
    nswd = sum(1 for s in spec if s["kind"] == E.TARGET_SWD)
    bytes_per_launch = nswd * B * (4 * L * 8 + K * 8 + 4)
    swd_ms_per_launch = fam_ms["swd"] / max(1, ncalls)
    achieved = bytes_per_launch / (swd_ms_per_launch * 1e-3) / 1e9
    lean = eng.last_swd_kernel() == "lean"
    # (committed counter passes: "<workload>" = the reference's sequence, "<workload>fast" = the engine's defaults -- short refinement
    #  + fast arithmetic = the trial-per-lane kernel --, "<workload>fastexact" = short refinement with the reference's arithmetic)
    pmc = pmc_summary(workload + ("" if eng.swd_search() == "reference" else ("fast" if lean else "fastexact")), B)
    # measured HBM bytes per launch (PMC pass of tools/profile_round.sh, committed summary); null without one
    traffic = pmc.get("hbm_bytes_per_launch")
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": pmc.get("source"),
            **pmc_stamp(pmc, swd_ms_per_launch),
            "traffic_over_algorithmic": (traffic / bytes_per_launch) if traffic else None,
            "traffic_note": ("below the algorithmic bytes: those count the model arrays once per target, the second target finds them in L2 (the models "
                             "are ordered inside eight blocks of the batch, a block per XCD: every L2 fetches an eighth of the arrays)" if lean else
                             "what exceeds the algorithmic bytes is not model data: the wavefronts' progress board (a word per hardware "
                             "wavefront slot, polled every 8 rounds) -- ~10 GB/s, a thousandth of the HBM peak"),
            "kernel": {"lean": "swd_lean_kernel (one lane per trial velocity; all dispersion targets of a step in one launch)",
                       "lane": "swd_kernel (one lane per evaluation; one launch per dispersion target)"}.get(
                           eng.last_swd_kernel(), "swd_group_kernel (all dispersion targets of a step in one launch)"),
            "kernel_ms_per_launch": swd_ms_per_launch, "algorithmic_bytes_per_launch": bytes_per_launch,
            "note": "required HBM line; the kernel is a scalar FP64 recurrence and is bound by FP64 vector issue, not by HBM "
                    "(SURVEY.md 8(d)): see binding"}
    if counters is not None:
        # flop model of SURVEY.md 8(d): layer-propagator steps, counted in the kernel PER WAVE TYPE, x flop-equivalents
        ev_r, ev_l, lps_r, lps_l = counters[8], counters[9], counters[10], counters[11]
        flop = lps_r * FLOP_PER_LPS[2] + lps_l * FLOP_PER_LPS[1]
        tf = flop / (swd_ms_per_launch * 1e-3) / 1e12
        roof["binding"] = {"bound": "fp64_valu", "achieved": tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s (flop-equivalents)",
                           "frac": tf / FP64_VALU_PEAK_TF,
                           "secular_evals_per_step": {"rayleigh": ev_r, "love": ev_l},
                           "layer_steps_per_step": {"rayleigh": lps_r, "love": lps_l},
                           "flop_equivalents_per_layer_step": {"rayleigh": FLOP_PER_LPS[2], "love": FLOP_PER_LPS[1]},
                           "valu_busy": pmc.get("valu_busy"), "active_lane_frac": pmc.get("active_lane_frac"),
                           "valu_insts_per_launch": pmc.get("valu_insts_per_launch"),
                           "note": "valu_busy = SQ_ACTIVE_INST_VALU x 4 cycles / (kernel time x 1024 SIMDs x clock), active_lane_frac = "
                                   "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): separate rocprofv3 --pmc pass "
                                   "(profiles/pmc_summary.json); null without one"}
    out = {
        "metric": "forward-model+logL evals/sec (batched 10-layer models)",
        "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" if not dryrun else "synthetic (DRY RUN: all ranks on one GPU, gloo)",
        "config": {"workload": {"c2": "joint Rayleigh+Love phase dispersion, %d-layer, 30 periods, batch=%d models/step/GPU, nocorr law" % (L, B),
                                "c2p": "c2's targets on models drawn from the chains' prior: ragged 2..%d layers, velocities in any order, thickmin 0.1, "
                                       "batch=%d models/step/GPU" % (L, B),
                                "c3": "joint Rayleigh+Love phase dispersion + P-RF (gauss 2.5, nsamp 2048), %d-layer, batch=%d models/step/GPU, exp law on RF" % (L, B),
                                "c2g": "joint Rayleigh+Love GROUP dispersion, %d-layer, 30 periods, batch=%d models/step/GPU, nocorr law" % (L, B),
                                "c3g": "c3 (%d-layer, batch=%d) with the Gauss law (fixed r = 0.92, rcond 1e-6) on the RF" % (L, B)}[workload],
                   "batch_per_gpu": B, "layers": L, "periods": int(K), "targets": [s["name"] for s in spec],
                   "search": eng.swd_search(),
                   "parallelism": "models sharded one batch per GPU, no data-path collective"},
        "ms_per_step_stats": {"min": float(per_step.min()), "median": float(np.median(per_step)), "max": float(per_step.max()),
                              "source": "the engine's HIP events on the launch stream: start of a step -> start of the next (last: its own span), rank 0"},
        "clock_warmup": {"ms": CLOCK_WARMUP_MS, "steps": n_clock, "note": "untimed launches of the same step before the --warmup steps"},
        "roofline": roof,
        "kernel_ms_per_step": {k: v / max(1, ncalls) for k, v in fam_ms.items()},
        "gpu_ms_per_step": tot_ms / max(1, ncalls),
        "failed_models_last_step": n_failed, "guarded_models_last_step": n_guarded, "logL_finite": finite,
    }
    if world > 1:
        out["rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms}
    if not args.no_parity:
        try:
            out["parity_check"] = parity_check(spec, batches[(args.warmup + args.steps - 1) % NPOOL], noise, d_logL, d_misf, d_err)
            # the synthetics themselves (dispersion velocities) against the oracle's REFERENCE sequence -- north_star's own
            # quantity and tolerance (1e-5 relative): one more call of the last step with the synthetics asked for, untimed
            d_ymod = torch.zeros((B, eng.ldy), dtype=torch.float64, device=dev)
            nl, h_, vp_, vs_, rho_ = d_batches[(args.warmup + args.steps - 1) % NPOOL]
            eng.evaluate_batch_dev(B, L, nl.data_ptr(), h_.data_ptr(), vp_.data_ptr(), vs_.data_ptr(), rho_.data_ptr(), B, 1,
                                   d_noise.data_ptr(), d_logL.data_ptr(), d_misf.data_ptr(), d_err.data_ptr(), ymod=d_ymod.data_ptr(),
                                   stream=stream)
            torch.cuda.synchronize()
            out["parity_check"].update(synthetics_check(spec, batches[(args.warmup + args.steps - 1) % NPOOL], noise, d_ymod, d_logL, d_err))
        except Exception as ex:
            out["parity_check"] = {"n": 0, "error": repr(ex)}
    if any(s["kind"] == E.TARGET_RF for s in spec) and args.full and not args.no_rf_roofline and rf_roof:
        try:
            out["rf_roofline"] = rf_roofline(eng, spec, d_batches[0], B, L, dev)
        except Exception as ex:
            out["rf_roofline"] = {"error": repr(ex)}
    if with_cpu and not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0 at N = 1 only
        t_cpu = time.perf_counter()
        have_ref = False
        try:   # the reference's own compiled code, where oracle/_ref was built (it travels with the repo)
            from oracle import refshim
            have_ref = refshim.available() and workload in ("c2", "c3", "c2g")
        except Exception:
            pass
        if args.full or not have_ref:        # the oracle port: the baseline where oracle/_ref is absent, beside it with --full
            try:
                out["cpu_baseline"] = cpu_baseline(spec, batches[0], noise, workload)
            except Exception as ex:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "evals/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (ex,)}
        if have_ref:
            try:
                # north_star: "next to the reference CPU Fortran path timed on the same box's host cores":
                # the reference's own compiled code is the baseline, the port is kept beside it (--full)
                ref = cpu_baseline_reference(spec, batches[0], noise, workload, (16, 32, 64, 128, os.cpu_count() or 1) if args.full else None)
                if "cpu_baseline" in out:
                    out["cpu_baseline_port"] = out["cpu_baseline"]
                out["cpu_baseline"] = ref
            except Exception as ex:
                out["cpu_baseline_reference_error"] = repr(ex)
                if "cpu_baseline" not in out:
                    try:
                        out["cpu_baseline"] = cpu_baseline(spec, batches[0], noise, workload)
                    except Exception as ex2:
                        out["cpu_baseline"] = {"value": None, "unit": "evals/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex2,)}
        out["cpu_baseline_wall_s"] = time.perf_counter() - t_cpu
    return out


def make_summary(out):
    """Every workload's [value, ms per step], c3 / c2, both roofline fractions: a few hundred characters."""
    def vm(b, nd=4):
        if not isinstance(b, dict) or b.get("value") is None:
            return None
        return [float("%.*g" % (nd, b["value"])), float("%.4g" % b["ms_per_step"])]
    sm = {"c2": vm(out)}
    for w in ("c3", "c2g", "c3g", "c4", "c5", "c5_full"):
        if w in out:
            sm[w] = vm(out[w])
    if isinstance(out.get("c2p"), dict) and out["c2p"].get("value") is not None:   # [value, ms/step, guarded fraction of the last step]
        sm["c2p"] = vm(out["c2p"]) + [float("%.3g" % (out["c2p"].get("guarded_models_last_step", 0) / max(1, out["c2p"]["config"]["batch_per_gpu"])))]
    for name, tag in (("reference_search", "_ref"), ("fast_search", "_fast")):
        fs = out.get(name, {})
        for w in ("c2", "c3", "c4", "c5", "c5_full"):
            if isinstance(fs.get(w), dict) and fs[w].get("value") is not None:
                sm[w + tag] = float("%.4g" % fs[w]["value"])
    if isinstance(out.get("c3"), dict) and "ratio_to_c2_ms_per_step" in out["c3"]:
        sm["c3/c2"] = float("%.4g" % out["c3"]["ratio_to_c2_ms_per_step"])
    sm["frac_hbm"] = float("%.3g" % out["roofline"]["frac"])
    if "binding" in out["roofline"]:
        sm["frac_fp64"] = float("%.3g" % out["roofline"]["binding"]["frac"])
    rf = out.get("c3", {}).get("rf_roofline") if isinstance(out.get("c3"), dict) else None
    if isinstance(rf, dict) and "frac" in rf:
        sm["rf_frac_hbm"] = float("%.3g" % rf["frac"])
        sm["rf_frac_fp64"] = float("%.3g" % rf["binding"]["frac"])
    for w in ("c5", "c5_full"):
        if isinstance(out.get(w), dict) and "config" in out[w]:
            sm[w + "_swaps"] = out[w]["config"].get("accepted_swaps")
    sm["search"] = out["config"].get("search")
    sm["units"] = "[value, ms/step]; evals/s (c2..c3g), chain-it/s (c4..); c2p: + guarded fraction"
    return sm


LINE_LIMIT = 8000      # the driver parses the line out of an 8 KB tail of stdout (BENCH_r04: a 24.6 KB line was not parsed)


def _sig(x, nd=6):
    """Numbers to `nd` significant digits (the line is a record, not an archive: the archive is bench_full.json)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (float, np.floating)):
        return float("%.*g" % (nd, float(x))) if np.isfinite(x) else None
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, dict):
        return {k: _sig(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, nd) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full, full_path=None):
    """The ONE line rank 0 prints: the contract's keys, `roofline` (+ binding), `cpu_baseline`, a five-key `parity_check`
    and the summary -- a whitelist, so that whatever the blocks grow by lands in the full record and never in the line."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data", "error", "failed_models_last_step", "guarded_models_last_step"))
    cfg = full.get("config", {})
    line["config"] = _pick(cfg, ("workload", "batch_per_gpu", "layers", "periods", "targets", "search", "parallelism",
                                 "chains_per_gpu", "accepted_swaps"))
    roof = full.get("roofline")
    if isinstance(roof, dict):
        r = _pick(roof, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel",
                         "kernel_ms_per_launch", "algorithmic_bytes_per_launch", "pmc_stale"))
        if isinstance(roof.get("binding"), dict):
            r["binding"] = _pick(roof["binding"], ("bound", "achieved", "peak", "unit", "frac", "valu_busy", "active_lane_frac"))
        line["roofline"] = r
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind"))
        c["sample"] = str(cb.get("sample", ""))[:200]
        line["cpu_baseline"] = c
    pc = full.get("parity_check")
    if isinstance(pc, dict):
        line["parity_check"] = _pick(pc, ("n", "failure_flags_equal", "max_rel_velocity", "velocity_tolerance",
                                          "max_rel_logL_of_the_device_synthetics", "max_abs_logL", "error"))
    for k in ("kernel_ms_per_step", "rank_ms_per_step", "collective_check", "speculation"):
        if k in full:
            v = full[k]
            if k == "rank_ms_per_step":
                v = _pick(v, ("min", "max"))
            if k == "speculation":
                v = _pick(v, ("depth", "evaluation_launches", "iterations_per_launch", "ms_per_launch"))
            line[k] = v
    if full_path:
        line["full_record"] = full_path
    if "summary" in full:
        line["summary"] = full["summary"]
    line = _sig(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:      # cannot happen with the whitelist above; if it ever does, the contract's keys survive
        for k in ("summary", "kernel_ms_per_step", "speculation", "parity_check"):
            line.pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="all", choices=["all", "c2", "c2p", "c3", "c2g", "c3g", "c4", "c5", "c5_full"],
                    help="all (default): the c2 line (headline) with the other configs in its summary; or one workload")
    ap.add_argument("--chains", type=int, default=0, help="c4/c5: chains per GPU (default 8 / 64)")
    ap.add_argument("--chain-steps", type=int, default=0, help="c4/c5: timed iterations per chain (default: --steps with "
                    "--workload c4/c5, 600 inside --workload all)")
    ap.add_argument("--spec-depth", type=int, default=0, help="c4/c5: iterations per evaluation launch (0 = automatic)")
    ap.add_argument("--search", default=None, choices=["reference", "fast", "fast_rayleigh"],
                    help="root refinement of the dispersion search (bh_engine_set_swd_search): fast = the engine's default, what "
                         "`value` is measured with; reference = the reference's own sequence, bit-identical velocities")
    ap.add_argument("--arith", default=None, choices=["fast", "exact"],
                    help="arithmetic of the launches in which every target takes the short refinement (bh_engine_set_swd_arith): "
                         "fast = the engine's default, what `value` is measured with (the trial-per-lane kernel); exact = the "
                         "reference's rounding points.  The chain workloads take DeviceChains' own default (fast as well) unless given")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=10)
    ap.add_argument("--full", action="store_true", help="also: the other search, the RF kernels alone, the port baseline, the CPU pool sweep")
    ap.add_argument("--out", default=os.path.join(REPO, "bench_full.json"), help="where the full record of every block is written")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rf-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the last step (after the timed region)")
    args = ap.parse_args()
    chain_search = args.search or "fast"   # (DeviceChains' own default)
    args.search = args.search or "fast"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (`python bench.py --gpus N`): become the launcher -- one rank per GPU under
        # torch.distributed.run on this node; rank 0 of the re-executed script prints the one JSON line
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    # stdout carries ONE line, the JSON record: everything libraries print there on the way (RCCL's version banner at
    # the first collective, gloo's connection messages) goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path in the product")
    # Dry-run aid for boxes with ONE GPU: BH_BENCH_DRYRUN=1 puts every rank on GPU 0 and uses gloo, so
    # that the N > 1 control flow (rendezvous, barriers, max-over-ranks) can be exercised; never set by
    # the driver, and the JSON line then says so in "data".
    dryrun = os.environ.get("BH_BENCH_DRYRUN", "0") == "1"
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        try:   # (first contact with RCCL must not be able to hang the driver's scaling run: two minutes, then a JSON error line)
            if dryrun:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
            else:
                dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=120))
        except Exception as ex:
            if rank == 0:
                os.dup2(real_stdout, 1)
                print(json.dumps({"metric": "forward-model+logL evals/sec (batched 10-layer models)", "value": None, "unit": "evals/s",
                                  "n_gpus": world, "error": "init_process_group failed: %r" % (ex,)}), flush=True)
            raise SystemExit(3)
        # first contact with the collective library, before anything is timed: every rank contributes 1
        one = torch.ones(1, dtype=torch.float64, device="cpu" if dryrun else dev)
        dist.all_reduce(one)
        comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen_by_all_reduce": int(one.item()),
                "rccl_ranks": int(one.item()) if dist.get_backend() == "nccl" else 0, "devices": torch.cuda.device_count()}

    from bayhunter_amd import engine as E
    eng = E.Engine(local_rank)
    eng.set_swd_search(args.search)
    if args.arith:
        eng.set_swd_arith(args.arith)
    t_start = time.perf_counter()
    out = None
    if args.workload in ("c4", "c5", "c5_full"):
        eng.set_swd_search(chain_search)   # (not given: the chains' own default, see below)
        out = run_chains(args, eng, rank, world, dist, dev, args.workload, args.chain_steps or args.steps, args.warmup)
    elif args.workload != "all":
        out = run_eval(args, eng, rank, world, dist, dev, args.workload, dryrun)
    else:
        # the headline (c2, the configuration the metric is quoted on); the other BASELINE configs as blocks of the full
        # record and [value, ms/step] pairs of the line's summary: c3 = configs[2] (same --steps / --warmup, its own timed
        # region), c4 / c5 = configs[3] / [4] per-GPU shares
        out = run_eval(args, eng, rank, world, dist, dev, "c2", dryrun)
        blocks = {}
        for w in ("c3", "c2g", "c3g", "c2p"):       # configs[2], the "second runs" of SURVEY.md 8(d), c2 on prior-like models (no CPU leg)
            try:
                blocks[w] = run_eval(args, eng, rank, world, dist, dev, w, dryrun, with_cpu=False, rf_roof=(w == "c3"))
            except Exception as ex:
                blocks[w] = {"error": repr(ex)}
        csteps = args.chain_steps or 600
        chain_workloads = ("c4", "c5") + (("c5_full",) if world == 1 else ())
        eng.set_swd_search(chain_search)
        try:
            for w in chain_workloads:
                try:
                    blocks[w] = run_chains(args, eng, rank, world, dist, dev, w, csteps, max(100, csteps // 2))
                except Exception as ex:          # the chain blocks must never take the headline number down with them
                    blocks[w] = {"error": repr(ex)}
        finally:
            eng.set_swd_search(args.search)
        # --full: the same workloads with the OTHER root refinement, reported beside and never as `value`
        alt, alt_name = {}, ("reference" if args.search != "reference" else "fast")
        if args.full and (world == 1 or os.environ.get("BH_BENCH_ALT_BLOCK", "0") == "1"):   # (a supplement: at N = 1 only)
            try:
                eng.set_swd_search(alt_name)
                for w in ("c2", "c3"):
                    g0 = np.array(eng.guard_totals())
                    alt[w] = run_eval(args, eng, rank, world, dist, dev, w, dryrun, with_cpu=False, rf_roof=False)
                    if alt[w] is not None and alt_name != "reference":
                        alt[w]["models_rerun_by_the_guard"] = int((np.array(eng.guard_totals()) - g0).sum())
                eng.set_swd_search(alt_name)
                for w in chain_workloads:
                    try:
                        alt[w] = run_chains(args, eng, rank, world, dist, dev, w, csteps, max(100, csteps // 2))
                    except Exception as ex:
                        alt[w] = {"error": repr(ex)}
            except Exception as ex:
                alt["error"] = repr(ex)
            finally:
                eng.set_swd_search(args.search)
        if rank == 0:
            c3 = blocks["c3"]
            if isinstance(c3, dict) and c3.get("ms_per_step"):
                c3["ratio_to_c2_ms_per_step"] = c3["ms_per_step"] / out["ms_per_step"]
            out.update(blocks)
            if alt:
                out[alt_name + "_search"] = alt
            out["summary"] = make_summary(out)
    if rank == 0 and out is not None:
        if comm is not None:
            out["collective_check"] = comm
        out["bench_wall_s"] = time.perf_counter() - t_start
        full_path = None
        try:
            with open(args.out, "w") as f:
                json.dump(out, f, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else repr(o))
            full_path = os.path.relpath(args.out, REPO) if os.path.abspath(args.out).startswith(REPO) else args.out
        except OSError:
            pass
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(compact_line(out, full_path), flush=True)
        os.dup2(2, 1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
