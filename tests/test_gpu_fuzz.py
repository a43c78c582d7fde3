"""Seed-varying randomised parity sweep of the engine's DEFAULT dispersion path (short refinement + fast arithmetic: the
trial-per-lane kernel) inside the suite (VERDICT r05 "Next" #6).

The default path has no bit-level restatement: it is held to the reference by tolerance (velocities within 2e-6, north_star
1e-5) and by identity of the failure flags and zero rows.  The fixed-seed tests (test_gpu_swd_lean.py) cover 1.7 million models;
this test draws ~200 000 NEW models on every tree: the seed is BH_FUZZ_SEED if set, else derived from the commit (`git rev-parse
HEAD` where a work tree is present) or -- on the GPU box, whose snapshot has no .git -- from the bytes of the built library, so a
new build walks new models, and a failure is reproduced with the printed seed.  Checked against the ORACLE's reference sequence
(oracle/swd_oracle.c search mode 0, bit-identical to the compiled reference): failure flags, zero rows, 2.5e-6 -- for bh_swd_batch on
sorted-velocity models with a low-velocity zone and on models drawn from a sampler's prior, and for the fused call's failure
pattern."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def fuzz_seed():
    env = os.environ.get("BH_FUZZ_SEED")
    if env:
        return int(env), "BH_FUZZ_SEED"
    try:
        head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=REPO, capture_output=True, text=True, timeout=10)
        if head.returncode == 0 and len(head.stdout.strip()) >= 8:
            return int(head.stdout.strip()[:8], 16), "git rev-parse HEAD"
    except Exception:
        pass
    from bayhunter_amd import engine as E
    return int(hashlib.sha256(open(E.LIB_PATH, "rb").read()).hexdigest()[:8], 16), "sha256 of libbh_engine.so"


def _configs(rs, total):
    """Batch shapes until `total` models are drawn: sizes on both sides of the trials-per-round thresholds, ragged and not."""
    n = 0
    while n < total:
        B = int(rs.choice([1, 7, 64, 65, 200, 700, 1500, 2600, 6000]))
        L = int(rs.choice([2, 3, 5, 8, 10, 13, 17, 21, 30]))
        K = int(rs.choice([1, 5, 21, 30, 60]))
        yield B, L, K
        n += B


def test_default_path_against_the_reference_sequence_on_fresh_models(engine, oracle, capsys):
    from bayhunter_amd.synth import synth_models, prior_models
    seed, source = fuzz_seed()
    rs = np.random.RandomState(seed % (2 ** 32))
    total = int(os.environ.get("BH_FUZZ_MODELS", "200000"))
    engine.set_swd_search("fast")
    engine.set_swd_arith("fast")
    worst, nmodels, nguard, nlean, jumps = 0.0, 0, 0, 0, []
    for B, L, K in _configs(rs, total):
        ragged = bool(rs.rand() < 0.6) and L > 2
        prior = bool(rs.rand() < 0.5) and L >= 2
        if prior:
            nlay, h, vp, vs, rho = prior_models(rs, B, L, nmin=2 if ragged else L)
        else:
            nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=float(rs.choice([0.0, 0.2, 0.5])), ragged=ragged)
            if L > 10:
                h[:-1] *= 10.0 / L
        per = np.sort(rs.uniform(1.0, 80.0, K)) if rs.rand() < 0.5 else np.linspace(2, 60, K)
        iwave, flsph = int(rs.choice([1, 2])), int(rs.rand() < 0.25)
        engine.set_swd_trials(int(rs.choice([0, 0, 0, 8, 16, 32])))       # by the call's shape, or pinned
        try:
            v, e = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0, flsph=flsph)
        finally:
            engine.set_swd_trials(0)
        nlean += int(engine.last_swd_kernel() == "lean")
        nguard += sum(engine.guard_stats()[0])
        rv, re_, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, 0, flsph=flsph)
        both = (v != 0) & (rv != 0)
        rel = np.zeros_like(v)
        rel[both] = np.abs(v[both] - rv[both]) / np.abs(rv[both])
        # a model on ANOTHER ROOT than the reference's (the known exception, below): further from the reference's value than two
        # end points of one root's brackets can be (2.3e-6, below) -- whether by 7e-6 (three sign changes within 2e-5 of a half-space
        # velocity: seed 3790146708) or by a third of the velocity --, or, further along the other mode's branch, with another
        # failure flag / zero row
        far = (rel.max(axis=1) > 2.5e-6) | (e != re_) | ((v == 0) != (rv == 0)).any(axis=1) if B else np.zeros(0, bool)
        jumps += [(prior, B, L, K, iwave, flsph, int(b), float(rel[b].max()), int(e[b]), int(re_[b])) for b in np.where(far)[0]]
        if (~far).any():
            worst = max(worst, float(rel[~far].max()))
        nmodels += B
    with capsys.disabled():
        print("\n[fuzz] seed %d (%s): %d models, worst relative difference %.3g, guarded %d, calls on the trial-per-lane kernel %d, "
              "models on another root than the reference's %d %s" % (seed, source, nmodels, worst, nguard, nlean, len(jumps), jumps[:4]))
    assert nlean > 0
    # north_star: 1e-5.  The fixed sets of test_gpu_swd_lean.py are held to 2e-6 (seen there: 1.4e-6); the bound is 2.3e-6 --
    # the root lies inside this path's final bracket (<= 1.3e-6 c wide) and inside the reference's (1e-6 c), either returns a point
    # of its own -- and fresh models have come to 1.95e-6.  A model beyond 2.5e-6 is therefore on another root: counted below.
    assert worst <= 2.5e-6, "seed %d: %.3g" % (seed, worst)
    # Failure flags and zero rows: the reference's -- with the known exception: a root within ~1e-6 c of a scan grid
    # point with a second root less than a step away (two modes that nearly touch, a channel mode's pole-zero pair): the cell that
    # holds both shows no sign change, so this grid sees a bracket where the
    # reference's -- 1e-6 c beside it -- walks past to another mode, or the other way round.  Measured on 9.2 million models drawn
    # from a sampler's prior (profiles/r06_fuzz_prior_final.txt): fewer than one in a million.  More than ONE such model among the
    # ~100 000 prior-like ones of this test is a bug; on the sorted-velocity models none has ever been seen.
    assert len(jumps) <= 1 and not [j for j in jumps if not j[0]], "seed %d: %s" % (seed, jumps)


def test_fused_call_failure_pattern_on_fresh_models(engine, oracle, capsys):
    """bh_evaluate_batch with the engine's defaults: which models fail (logL = -1e15) is the reference sequence's; the misfits
    of the others within 3e-5 absolute (velocities move by <= 2e-6 relative)."""
    from bayhunter_amd import engine as E
    from bayhunter_amd.synth import synth_models, prior_models
    seed, source = fuzz_seed()
    rs = np.random.RandomState((seed + 1) % (2 ** 32))
    engine.set_swd_search("fast")
    engine.set_swd_arith("fast")
    flagdiff, worst, nmodels = 0, 0.0, 0
    for it in range(24):
        B = int(rs.choice([9, 130, 400, 1700]))
        L = int(rs.choice([3, 6, 10, 15, 21]))
        if rs.rand() < 0.5:
            nlay, h, vp, vs, rho = prior_models(rs, B, L)
        else:
            nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=0.2, ragged=bool(rs.rand() < 0.5))
            if L > 10:
                h[:-1] *= 10.0 / L
        nt = int(rs.randint(1, 4))
        spec = []
        for t in range(nt):
            K = int(rs.choice([7, 21, 30]))
            per = np.linspace(2, 50, K)
            law = int(rs.choice([E.LAW_NOCORR, E.LAW_EXP]))
            spec.append(dict(kind=E.TARGET_SWD, law=law, n=K, x=per, iwave=int(rs.choice([1, 2])), igr=0, yobs=3.0 + 0.02 * per + rs.normal(0, 0.02, K)))
        noise = np.zeros((B, 2 * nt))
        for t, s in enumerate(spec):
            noise[:, 2 * t] = rs.uniform(0.2, 0.9, B) if s["law"] == E.LAW_EXP else 0.0
            noise[:, 2 * t + 1] = rs.uniform(0.005, 0.1, B)
        engine.set_targets(spec)
        logL, misf, err = engine.evaluate_batch(nlay, h, vp, vs, noise)
        rL, rm = oracle.joint_batch(nlay, h.T, vp.T, vs.T, rho.T, spec, noise)
        flagdiff += int(((rL <= -1e14) != (logL <= -1e14)).sum())
        ok = (rL > -1e14) & (logL > -1e14)
        if ok.any():
            worst = max(worst, float(np.max(np.abs(misf[ok] - rm[ok]))))
        nmodels += B
    with capsys.disabled():
        print("\n[fuzz] fused call, seed %d (%s): %d models, failure patterns differing %d, worst misfit difference %.2e" %
              (seed + 1, source, nmodels, flagdiff, worst))
    assert flagdiff == 0, "seed %d" % (seed + 1)
    assert worst <= 3e-5, "seed %d: %.3g" % (seed + 1, worst)
