"""The short root refinement of the dispersion search (bh_engine_set_swd_search(e, BH_SEARCH_FAST)) with its guard.

Not the reference's sequence of evaluations, so the gate is north_star's tolerance, stated here:
    RTOL = 1e-5 relative on the dispersion velocities; failure flags and the period from which a failed model's row is
    zero IDENTICAL to the reference's (the guard sends the models whose outcome hinges on the last bits of a root back to
    the reference's sequence, swd_common.h) -- asserted on 2.3 million LVZ-rich ragged models below
against the oracle's restatement of the reference (bit-identical to surfdisp96, tests/test_oracle_swd.py) and against
the reference's own golden vectors.  What is achieved (asserted below as ACHIEVED): 1.2e-6 -- the reference's own stop
test leaves its root known to 1e-6 relative, the short refinement to 5e-8.
Second, the device's sequence in this mode is checked BIT FOR BIT, evaluation for evaluation, against a CPU restatement
of it (oracle/swd_oracle.c: search mode 2 + scan mode 1), and for independence of the launch plan."""
import numpy as np
import pytest

from conftest import golden
from bayhunter_amd.synth import synth_models

pytestmark = pytest.mark.gpu
REFS = {"rdispph": (2, 0), "rdispgr": (2, 1), "ldispph": (1, 0), "ldispgr": (1, 1)}
RTOL = 1e-5       # north_star
ACHIEVED = 1.2e-6


@pytest.fixture()
def fast(engine):
    """the short refinement; the counted scan wherever a Love target is, as the oracle's restatement has it (evaluation
    counts are compared; velocities and flags do not depend on the scan setting)"""
    engine.set_swd_search("fast")
    engine.set_swd_scan("counted")
    try:
        yield engine
    finally:
        engine.set_swd_search("reference")
        engine.set_swd_scan("auto")


class restatement:
    """the oracle's restatement of what the engine runs in this mode: guarded short refinement + counted scan"""

    def __init__(self, oracle):
        self.O = oracle

    def __enter__(self):
        self.O.set_swd_search(2)
        self.O.lib().bho_swd_set_scan(1)

    def __exit__(self, *a):
        self.O.set_swd_search(0)
        self.O.lib().bho_swd_set_scan(0)


def worst_rel(v, ov, ok):
    return float(np.max(np.abs(v[ok] - ov[ok]) / np.abs(ov[ok]))) if ok.any() else 0.0


@pytest.mark.parametrize("ref", ["rdispph", "ldispph"])
def test_within_tolerance_of_the_reference_sequence_lvz_rich(fast, oracle, ref):
    """20k models of 2..12 layers, a quarter with a low-velocity layer (the parity-statistics set of test_gpu_swd.py)."""
    rs = np.random.RandomState(2024)
    nlay, h, vp, vs, rho = synth_models(rs, 20000, 12, lvz_frac=0.25, ragged=True)
    per = np.linspace(2, 60, 30)
    iwave, igr = REFS[ref]
    fast.set_instrumentation(False, True)
    try:
        v, e = fast.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)
        nfast = fast.last_neval()
    finally:
        fast.set_instrumentation(False, False)
    ov, oe, nref = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr)
    assert np.array_equal(e, oe)                       # the same models fail
    both = (v != 0) & (ov != 0)
    w = worst_rel(v, ov, both)
    assert w <= RTOL, w
    assert w <= ACHIEVED, w
    assert np.array_equal(v == 0, ov == 0)             # and their rows are zero from the same period on
    assert nfast < 0.8 * nref                          # and it is shorter: the point of the mode


BIG = [  # (target, layers up to, higher modes, earth flattening, models): 2.3 million in all
    ("ldispph", 4, 1, 0, 500000), ("ldispph", 12, 1, 0, 400000), ("ldispph", 21, 2, 0, 200000), ("ldispph", 8, 1, 1, 200000),
    ("rdispph", 12, 1, 0, 400000), ("rdispph", 6, 2, 0, 200000), ("rdispph", 8, 1, 1, 200000),
    ("ldispgr", 12, 1, 0, 100000), ("rdispgr", 12, 2, 0, 100000)]


@pytest.mark.parametrize("ref,L,mode,flsph,n", BIG)
def test_failure_flags_and_zero_rows_are_the_references_on_millions_of_models(fast, oracle, ref, L, mode, flsph, n):
    """LVZ-rich ragged models, all four target kinds, modes 1-2 (the short refinement is for fundamental-mode
    phase velocities; the others must come out as the reference's bits), flat and flattened earth -- 0 differing
    failure flags, 0 rows whose zero pattern differs, velocities within 1e-5 (achieved 1.2e-6) of the oracle's restatement of
    the REFERENCE's sequence (group velocities: its bits).  Thin models observed out to 60 s are where a Love root creeps up
    to the half-space velocity and the unguarded short sequence differed."""
    iwave, igr = REFS[ref]
    per = np.linspace(2, 60, 30)
    rs = np.random.RandomState(1000 + L * 7 + mode * 3 + flsph + iwave * 100 + igr * 1000)
    nguard = 0
    for _ in range(n // 50000):
        nlay, h, vp, vs, rho = synth_models(rs, 50000, L, lvz_frac=0.25, ragged=True)
        v, e = fast.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr, mode=mode, flsph=flsph)
        nguard += sum(fast.guard_stats()[0])
        ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr, mode=mode, flsph=flsph)
        assert np.array_equal(e, oe)
        assert np.array_equal(v == 0, ov == 0)
        if igr == 1 or mode > 1:      # group velocities and targets with higher modes keep the reference sequence: its bits
            assert np.array_equal(v, ov)
        else:
            assert worst_rel(v, ov, (v != 0) & (ov != 0)) <= ACHIEVED
    if igr == 1 or mode > 1:
        assert nguard == 0
    elif iwave == 1:
        assert nguard > 0      # (the sets do contain the situations the guard is there for)


def test_counted_scan_keeps_the_bits_and_matches_its_restatement_evaluation_for_evaluation(engine, oracle):
    """Reference search, Love: BH_SCAN_COUNTED (default) against BH_SCAN_STEPS -- identical velocities and flags, fewer
    evaluations; each against the oracle's restatement of the same scan, count for count.  Rayleigh: no difference."""
    rs = np.random.RandomState(31)
    nlay, h, vp, vs, rho = synth_models(rs, 6000, 10, lvz_frac=0.3, ragged=True)
    per = np.sort(rs.uniform(1.0, 80.0, 30))   # (irregular periods, higher modes, group velocities: where a mode's start lies below its floor)
    a = [np.ascontiguousarray(x.T) for x in (h, vp, vs, rho)]
    assert engine.swd_scan() == "auto"
    engine.set_instrumentation(False, True)
    try:
        for iwave, igr, mode, flsph in ((1, 0, 1, 0), (1, 1, 2, 0), (1, 0, 3, 1), (1, 1, 3, 0), (2, 0, 1, 0)):
            got = {}
            for scan in ("counted", "steps"):
                engine.set_swd_scan(scan)
                got[scan] = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr, mode=mode, flsph=flsph) + (engine.last_neval(),)
            ov, oe, n0 = oracle.swd_batch(nlay, *a, per, iwave, igr, mode=mode, flsph=flsph)
            with oracle.swd_scan(1):
                cv, ce, n1 = oracle.swd_batch(nlay, *a, per, iwave, igr, mode=mode, flsph=flsph)
            for scan, n in (("counted", n1), ("steps", n0)):
                assert np.array_equal(got[scan][0], ov) and np.array_equal(got[scan][1], oe), (iwave, igr, mode, scan)
                assert got[scan][2] == n, (iwave, igr, mode, scan, got[scan][2], n)
            assert np.array_equal(cv, ov) and np.array_equal(ce, oe)
            assert (n1 < 0.8 * n0) if iwave == 1 else (n1 == n0)
    finally:
        engine.set_swd_scan("auto")
        engine.set_instrumentation(False, False)


@pytest.mark.parametrize("ref", sorted(REFS))
def test_device_sequence_equals_its_cpu_restatement(fast, oracle, ref):
    rs = np.random.RandomState(101)
    nlay, h, vp, vs, rho = synth_models(rs, 777, 21, lvz_frac=0.25, ragged=True)
    vs[0, :8] = 0.0                      # a few models with a water layer on top
    vp[0, :8] = 1.5
    vs[:, 8:12] *= 0.2                   # and some that fail the search
    per = np.linspace(2, 60, 30)
    iwave, igr = REFS[ref]
    for mode in (1, 2):
        # (evaluation for evaluation: with a group velocity's two chains of roots in ONE launch, as the restatement runs them --
        #  as two launches the counted Love scan of a second root has no previous stride to start from: same bits, other count)
        fast.set_instrumentation(False, True)
        fast.set_tuning("swd_gsplit", 0)
        try:
            v, e = fast.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr, mode=mode)
            n = fast.last_neval()
        finally:
            fast.set_tuning("swd_gsplit", 1 << 24)
            fast.set_instrumentation(False, False)
        v2, e2 = fast.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr, mode=mode)
        assert np.array_equal(v2, v) and np.array_equal(e2, e), (ref, mode)
        with restatement(oracle):
            ov, oe, on = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr, mode=mode)
        assert np.array_equal(e, oe) and np.array_equal(v, ov), (ref, mode)
        assert n == on                   # evaluation for evaluation
        if igr == 1:                     # group-velocity targets keep the reference sequence: the reference's bits
            with oracle.swd_scan(1):
                rv, re_, rn = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr, mode=mode)
            assert np.array_equal(v, rv) and np.array_equal(e, re_) and n == rn


def test_reference_golden_vectors_within_tolerance(fast):
    g = golden("swd_golden.npz")
    nlay = g["nlay"]
    for pset in ("p21", "p30"):
        per = g["x_" + pset]
        for ir, ref in enumerate(g["refs"]):
            iwave, igr = REFS[str(ref)]
            vel, err = fast.swd_batch(nlay, g["h"], g["vp"], g["vs"], g["rho"], per, iwave, igr, layout="model_major")
            ok = g["ok_" + pset][:, ir].astype(bool)
            assert np.array_equal(err == 0, ok)
            ref_y = g["y_" + pset][:, ir]
            assert worst_rel(vel, ref_y, ok) <= RTOL
            if igr == 1:
                assert np.array_equal(vel[ok], ref_y[ok])


@pytest.mark.parametrize("G,J", [(1, 1), (1, 4), (1, 16), (5, 2), (9, 1), (9, 3), (9, 7), (16, 2), (21, 3)])
def test_result_does_not_depend_on_the_launch_plan(fast, G, J):
    """One lane per model or G lanes per model, any look-ahead, alone or in a batch: the same bits and the same number
    of consumed evaluations -- the sequence is a function of the model."""
    rs = np.random.RandomState(77)
    nlay, h, vp, vs, rho = synth_models(rs, 200, 12, lvz_frac=0.3, ragged=True)
    vs[:, 8:12] *= 0.2
    per = np.linspace(1.5, 70, 35)
    try:
        for iwave in (2, 1):
            for mode in (1, 3):
                fast.set_swd_group(9)
                fast.set_swd_lookahead(2)
                fast.set_instrumentation(False, True)
                v1, e1 = fast.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0, mode=mode)
                n1 = fast.last_neval()
                fast.set_swd_group(G)
                fast.set_swd_lookahead(J)
                v2, e2 = fast.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0, mode=mode)
                n2 = fast.last_neval()
                assert np.array_equal(v1, v2) and np.array_equal(e1, e2), (iwave, mode)
                assert n1 == n2 and n1 > 0
        fast.set_swd_group(0)
        fast.set_swd_lookahead(0)
        for b in (0, 17, 199):           # a model alone (the planner's latency mapping) = the model inside the batch
            vb, eb = fast.swd_batch(nlay[b:b + 1], h[:, b:b + 1], vp[:, b:b + 1], vs[:, b:b + 1], rho[:, b:b + 1], per, 2, 0)
            va, ea = fast.swd_batch(nlay, h, vp, vs, rho, per, 2, 0)
            assert np.array_equal(vb[0], va[b]) and eb[0] == ea[b]
    finally:
        fast.set_swd_group(0)
        fast.set_swd_lookahead(0)
        fast.set_instrumentation(False, False)


def test_earth_flattening_and_deep_models(fast, oracle):
    rs = np.random.RandomState(9)
    per = np.linspace(2, 60, 30)
    for L, B, fl in ((12, 300, 1), (50, 40, 0), (100, 5, 0)):
        nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=0.2, ragged=(L == 12), hmin=0.5 if L > 12 else 1.5, hmax=2.0 if L > 12 else 8.0)
        for iwave in (2, 1):
            v, e = fast.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0, flsph=fl)
            ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, 0, flsph=fl)
            assert np.array_equal(e, oe)
            assert worst_rel(v, ov, oe == 0) <= ACHIEVED
            with restatement(oracle):
                fv, fe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, 0, flsph=fl)
            assert np.array_equal(v, fv) and np.array_equal(e, fe)


def test_broken_models_are_failed_in_band(fast, oracle):
    """NaN / infinite / negative / absurd parameters are reported as failed without being searched, the healthy models
    of the batch are unaffected (test_gpu_swd.py has the same for the reference sequence)."""
    rs = np.random.RandomState(8)
    nlay, h, vp, vs, rho = synth_models(rs, 24, 6)
    vs[2, 0] = np.nan; vp[1, 1] = np.inf; h[0, 2] = np.nan; rho[3, 3] = -1.0; vs[:, 4] = -1.0; h[1, 5] = -5.0
    vp[:, 6] = 1e-60; vs[:, 7] = 1e30
    per = np.linspace(2, 40, 12)
    good = np.arange(8, 24)
    try:
        for G, J in ((0, 0), (1, 4), (9, 2), (9, 7)):
            fast.set_swd_group(G)
            fast.set_swd_lookahead(J)
            for iwave in (2, 1):
                v, e = fast.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0)
                assert (e[:8] == 1).all() and (v[:8] == 0).all()
                with restatement(oracle):
                    ov, oe, _ = oracle.swd_batch(nlay[good], h.T[good], vp.T[good], vs.T[good], rho.T[good], per, iwave, 0)
                assert np.array_equal(v[good], ov) and np.array_equal(e[good], oe)
    finally:
        fast.set_swd_group(0)
        fast.set_swd_lookahead(0)


def test_speculative_chain_windows_stay_exact_with_the_short_refinement(fast):
    """The short sequence is a function of the model alone, so the device-resident chains' speculative windows (many
    proposals per launch, batch composition changing with the depth) still walk the one-iteration-per-launch trajectory
    bit for bit -- and it is a different trajectory from the reference sequence's (1e-6 in a velocity flips a decision
    sooner or later)."""
    from test_gpu_device_chains import SETUPS, make_targets
    from bayhunter_amd.device_chains import DeviceChains
    g = golden("chain_golden.npz")
    su = SETUPS["exp"]
    init = dict(su["init"], iter_burnin=420, iter_main=80, maxmodels=10)
    keys = ("n", "vs", "z", "like", "noise", "vpvs", "misfits", "propdist", "accepted")

    def run(depth):
        return DeviceChains(make_targets(g), 8, init, su["priors"], seed=11, spec_depth=depth, search=None).run().state_host()   # (the engine's setting)

    s1, s5 = run(1), run(5)
    for k in keys:
        assert np.array_equal(s1[k], s5[k]), k
    fast.set_swd_search("reference")
    r1 = run(1)
    fast.set_swd_search("fast")
    assert not np.array_equal(r1["like"], s1["like"])
    assert abs(np.median(r1["like"]) - np.median(s1["like"])) < 0.2 * abs(np.median(r1["like"])) + 50.0


def test_fast_rayleigh_keeps_the_reference_sequence_for_love_targets(engine, oracle):
    """BH_SEARCH_FAST_RAYLEIGH: in ONE launch (the build with both sequences) the Rayleigh phase target takes the guarded
    short refinement -- the bits of its restatement --, the Love phase target and the group-velocity target the reference's
    sequence -- the reference's bits; for several models per wavefront and for one model per wavefront."""
    from bayhunter_amd import engine as E
    per = np.linspace(2, 60, 30)
    rs = np.random.RandomState(41)
    try:
        engine.set_swd_search("fast_rayleigh")
        assert engine.swd_search() == "fast_rayleigh"
        for B in (300, 3000):
            nlay, h, vp, vs, rho = synth_models(rs, B, 12, lvz_frac=0.25, ragged=True)
            a = [np.ascontiguousarray(x.T) for x in (h, vp, vs, rho)]
            yobs = 3.4 + 0.01 * per
            engine.set_targets([dict(kind=E.TARGET_SWD, law=0, n=30, x=per, yobs=yobs, iwave=2, igr=0),
                                dict(kind=E.TARGET_SWD, law=0, n=30, x=per, yobs=yobs, iwave=1, igr=0),
                                dict(kind=E.TARGET_SWD, law=0, n=30, x=per, yobs=yobs, iwave=2, igr=1)])
            noise = np.tile([0, 0.05] * 3, (B, 1))
            logL, misf, err, ymod = engine.evaluate_batch(nlay, h, vp, vs, noise, want_ymod=True)
            with restatement(oracle):
                rv, re_, _ = oracle.swd_batch(nlay, *a, per, 2, 0)
            lv, le, _ = oracle.swd_batch(nlay, *a, per, 1, 0)
            gv, ge, _ = oracle.swd_batch(nlay, *a, per, 2, 1)
            ok = (re_ == 0) & (le == 0) & (ge == 0)
            assert np.array_equal(err == 0, ok)
            assert np.array_equal(ymod[ok, :30], rv[ok]) and np.array_equal(ymod[ok, 30:60], lv[ok]) and np.array_equal(ymod[ok, 60:], gv[ok])
            # and target by target (a Love-only call takes the reference build, a Rayleigh-only call the short one)
            v, e = engine.swd_batch(nlay, h, vp, vs, rho, per, 1, 0)
            assert np.array_equal(v, lv) and np.array_equal(e, le)
            v, e = engine.swd_batch(nlay, h, vp, vs, rho, per, 2, 0)
            assert np.array_equal(v, rv) and np.array_equal(e, re_)
    finally:
        engine.set_swd_search("reference")


def test_joint_launch_with_the_default_search_matches_the_restatement_whatever_the_scan(engine, oracle):
    """The launch bench.py's headline times: Rayleigh + Love phase targets of a fused call in ONE launch with the short
    refinement, where BH_SCAN_AUTO gives the Love wavefronts the counted scan (the <.., FASTM = 2, .., CNTB> build).  Both
    targets' synthetics carry the bits of the CPU restatement under every scan setting, for batches that several models
    share a wavefront in (300, 4096) and LVZ-rich ragged models; failure flags the reference's."""
    from bayhunter_amd import engine as E
    per = np.linspace(2, 60, 30)
    yobs = 3.4 + 0.01 * per
    rs = np.random.RandomState(91)
    try:
        engine.set_swd_search("fast")
        for B, L, lvz in ((300, 10, 0.0), (4096, 10, 0.25), (2500, 12, 0.5)):
            nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=lvz, ragged=True)
            a = [np.ascontiguousarray(x.T) for x in (h, vp, vs, rho)]
            engine.set_targets([dict(kind=E.TARGET_SWD, law=0, n=30, x=per, yobs=yobs, iwave=2, igr=0),
                                dict(kind=E.TARGET_SWD, law=0, n=30, x=per, yobs=yobs, iwave=1, igr=0)])
            noise = np.tile([0, 0.05] * 2, (B, 1))
            with restatement(oracle):
                rv, re_, _ = oracle.swd_batch(nlay, *a, per, 2, 0)
                lv, le, _ = oracle.swd_batch(nlay, *a, per, 1, 0)
            fr = oracle.swd_batch(nlay, *a, per, 2, 0)[1] | oracle.swd_batch(nlay, *a, per, 1, 0)[1]
            ok = (re_ == 0) & (le == 0)
            assert np.array_equal(ok, fr == 0)
            first = None
            for scan in ("auto", "steps", "counted"):
                engine.set_swd_scan(scan)
                logL, misf, err, ymod = engine.evaluate_batch(nlay, h, vp, vs, noise, want_ymod=True)
                assert np.array_equal(err == 0, ok), (B, scan)
                assert np.array_equal(ymod[ok, :30], rv[ok]) and np.array_equal(ymod[ok, 30:], lv[ok]), (B, scan)
                first = logL if first is None else first
                assert np.array_equal(logL, first), (B, scan)
    finally:
        engine.set_swd_scan("auto")
        engine.set_swd_search("reference")


def test_guarded_models_restart_inside_their_own_wavefront_in_one_model_per_wavefront_launches(fast, oracle):
    """A launch of one model per wavefront (a sampler's window, a small batch) takes the build with both sequences; a model the
    guard fires on starts again with the reference's sequence in the same wavefront -- no re-run launch.  Thin models observed
    out to 60 s (the guard fires on a fifth of them): the rows and flags of the restatement (which runs such a model a second
    time with the reference's sequence), evaluation for evaluation; the guard statistics count the restarts."""
    rs = np.random.RandomState(733)
    per = np.linspace(2, 60, 30)
    for B, L in ((120, 4), (300, 6)):
        nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=0.2, ragged=True)
        a = [np.ascontiguousarray(x.T) for x in (h, vp, vs, rho)]
        for iwave in (1, 2):
            launches0 = fast.guard_stats()[1]
            fast.set_instrumentation(False, True)
            try:
                v, e = fast.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0)
                n = fast.last_neval()
            finally:
                fast.set_instrumentation(False, False)
            counts, launches = fast.guard_stats()[:2]
            oracle.swd_guarded_count(reset=True)
            with restatement(oracle):
                ov, oe, on = oracle.swd_batch(nlay, *a, per, iwave, 0)
            og = oracle.swd_guarded_count(reset=True)
            assert np.array_equal(v, ov) and np.array_equal(e, oe) and n == on, (B, L, iwave)
            assert sum(counts) == og, (counts, og)
            assert launches == launches0                    # in place: no re-run launch was enqueued
            if iwave == 1 and L == 4:
                assert og > 0                               # (the case is there to exercise the restart)


def test_default_search_is_the_short_refinement(engine, oracle):
    """A fresh engine takes the guarded short refinement for fundamental-mode phase velocities (with the reference's
    arithmetic selected: the bits of its CPU restatement; its own default, the fast arithmetic: tests/test_gpu_swd_lean.py) and
    the reference's sequence for everything else (group velocities, higher modes: the reference's bits); the settings are per
    engine."""
    from bayhunter_amd import engine as E
    eng = E.Engine(0)
    try:
        assert eng.swd_search() == "fast" and engine.swd_search() == "reference"
        assert eng.swd_arith() == "fast" and engine.swd_arith() == "exact"
        eng.set_swd_arith("exact")
        rs = np.random.RandomState(77)
        nlay, h, vp, vs, rho = synth_models(rs, 700, 10, lvz_frac=0.25, ragged=True)
        a = [np.ascontiguousarray(x.T) for x in (h, vp, vs, rho)]
        per = np.linspace(2, 60, 30)
        for iwave in (1, 2):
            v, e = eng.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0)
            with restatement(oracle):
                ov, oe, _ = oracle.swd_batch(nlay, *a, per, iwave, 0)
            rv, re_, _ = oracle.swd_batch(nlay, *a, per, iwave, 0)
            assert np.array_equal(v, ov) and np.array_equal(e, oe) and np.array_equal(e, re_)
            both = (v != 0) & (rv != 0)
            assert np.array_equal(v == 0, rv == 0) and np.max(np.abs(v[both] - rv[both]) / rv[both]) <= 2e-6
            for kw in (dict(igr=1), dict(igr=0, mode=2)):
                v, e = eng.swd_batch(nlay, h, vp, vs, rho, per, iwave, kw["igr"], mode=kw.get("mode", 1))
                rv, re_, _ = oracle.swd_batch(nlay, *a, per, iwave, kw["igr"], mode=kw.get("mode", 1))
                assert np.array_equal(v, rv) and np.array_equal(e, re_)
    finally:
        eng.close()


def test_the_switch_is_per_engine_and_validated(engine):
    from bayhunter_amd.engine import EngineError
    assert engine.swd_search() == "reference"          # (set by conftest for every gpu test; a fresh engine: see above)
    with pytest.raises(ValueError):
        engine.set_swd_search("quick")
    engine.set_swd_search("fast")
    assert engine.swd_search() == "fast"
    engine.set_swd_search("reference")
    rc = engine._L.bh_engine_set_swd_search(engine._h, 7)
    assert rc != 0
    with pytest.raises(EngineError):
        engine._check(rc)
