"""The engine's DEFAULT dispersion path: short refinement (BH_SEARCH_FAST) + fast arithmetic (BH_ARITH_FAST) = the
trial-per-lane kernel of csrc/swd_lean.hip (fundamental-mode phase velocities).

Neither the reference's sequence of evaluations nor its rounding points, so the gate is north_star's tolerance, stated here:
    RTOL = 1e-5 relative on the dispersion velocities; failure flags and the period from which a failed model's row is
    zero IDENTICAL to the reference's (the guard sends models whose outcome could hinge on the last bits of a root, or on
    the sign of a value the rounding error could decide, back to the reference's sequence in the reference's arithmetic)
against the oracle's restatement of the REFERENCE (bit-identical to surfdisp96, tests/test_oracle_swd.py) and against the
reference's own golden vectors.  ACHIEVED (asserted): 2e-6 -- the reference's own stop test leaves its root known to 1e-6
relative, both outputs are rounded to binary32 (6e-8 each), and this path accepts a bracket of at most 1.3e-6 (typically 2e-7)
and returns the interpolated point inside it (off by ~1e-10 where the function is smooth, by up to 4e-7 observed at a root
right below a half-space velocity, where it has a square-root kink)."""
import numpy as np
import pytest

from conftest import golden
from bayhunter_amd.synth import synth_models

pytestmark = pytest.mark.gpu
REFS = {"rdispph": (2, 0), "ldispph": (1, 0)}
RTOL = 1e-5       # north_star
ACHIEVED = 2.0e-6


@pytest.fixture()
def lean(engine):
    """the engine's default settings (conftest puts every gpu test on the reference's search and arithmetic, and back)"""
    engine.set_swd_search("fast")
    engine.set_swd_arith("fast")
    engine.set_swd_scan("auto")
    yield engine


def worst_rel(v, ov, ok):
    return float(np.max(np.abs(v[ok] - ov[ok]) / np.abs(ov[ok]))) if ok.any() else 0.0


def check_against_the_reference(v, e, ov, oe):
    assert np.array_equal(e, oe)                       # the same models fail
    assert np.array_equal(v == 0, ov == 0)             # and their rows are zero from the same period on
    w = worst_rel(v, ov, (v != 0) & (ov != 0))
    assert w <= RTOL, w
    assert w <= ACHIEVED, w
    return w


def test_a_fresh_engine_computes_this_way(oracle):
    from bayhunter_amd import engine as E
    eng = E.Engine(0)
    assert eng.swd_search() == "fast" and eng.swd_arith() == "fast"
    rs = np.random.RandomState(3)
    nlay, h, vp, vs, rho = synth_models(rs, 700, 10, lvz_frac=0.2)
    per = np.linspace(2, 60, 30)
    for iwave in (2, 1):
        v, e = eng.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0)
        assert eng.last_swd_kernel() == "lean"
        ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, 0)
        check_against_the_reference(v, e, ov, oe)
    # group velocities, higher modes, the reference's search or arithmetic: the other kernels
    eng.swd_batch(nlay, h, vp, vs, rho, per, 2, 1)
    assert eng.last_swd_kernel() != "lean"
    eng.swd_batch(nlay, h, vp, vs, rho, per, 2, 0, mode=2)
    assert eng.last_swd_kernel() != "lean"
    with eng.computing("exact"):
        eng.swd_batch(nlay, h, vp, vs, rho, per, 2, 0)
        assert eng.last_swd_kernel() != "lean"
    with eng.searching("reference"):
        v, e = eng.swd_batch(nlay, h, vp, vs, rho, per, 2, 0)
        assert eng.last_swd_kernel() != "lean"
        ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, 2, 0)
        assert np.array_equal(v, ov) and np.array_equal(e, oe)      # the reference's bits stay one call away


@pytest.mark.parametrize("ref", sorted(REFS))
def test_within_tolerance_of_the_reference_lvz_rich(lean, oracle, ref):
    """20k models of 2..12 layers, a quarter with a low-velocity layer (the parity-statistics set of test_gpu_swd.py): 8 trials
    per round (a call of 10 241 ... 28 672 models); the first 3000 again as a call of their own: 32 trials per round."""
    rs = np.random.RandomState(2024)
    nlay, h, vp, vs, rho = synth_models(rs, 20000, 12, lvz_frac=0.25, ragged=True)
    per = np.linspace(2, 60, 30)
    iwave, igr = REFS[ref]
    v, e = lean.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)
    assert lean.last_swd_kernel() == "lean"
    ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr)
    check_against_the_reference(v, e, ov, oe)
    n = 3000
    v16, e16 = lean.swd_batch(nlay[:n], h[:, :n], vp[:, :n], vs[:, :n], rho[:, :n], per, iwave, igr)
    check_against_the_reference(v16, e16, ov[:n], oe[:n])
    # between the two trial counts: the last bit of the binary32 output, except where one of them sends the model to the
    # reference's sequence and the other does not (the guard's tests look at the root's last digits) -- then the reference's 1e-6
    assert worst_rel(v16, v[:n], v16 != 0) <= ACHIEVED
    # (round 6: with 32 trials per round a cell that holds betmx or a half-space velocity is resolved inside the kernel, with 8
    # it goes to the reference's sequence -- Love's long periods sit next to the half-space velocity: a few per cent of the values)
    assert np.mean(v16 != v[:n]) < 0.06


BIG = [  # (target, layers up to, earth flattening, models): 1.7 million in all
    ("ldispph", 4, 0, 500000), ("ldispph", 12, 0, 400000), ("ldispph", 8, 1, 200000),
    ("rdispph", 12, 0, 400000), ("rdispph", 8, 1, 200000)]


@pytest.mark.parametrize("ref,L,flsph,n", BIG)
def test_failure_flags_and_zero_rows_are_the_references_on_millions_of_models(lean, oracle, ref, L, flsph, n):
    """The sets of tests/test_gpu_swd_fast.py (LVZ-rich ragged models, flat and flattened earth; thin models observed out to
    60 s are where a Love root creeps up to the half-space velocity): 0 differing failure flags, 0 rows whose zero pattern
    differs, velocities within 1e-5 (achieved 2e-6) of the oracle's restatement of the REFERENCE."""
    iwave, igr = REFS[ref]
    per = np.linspace(2, 60, 30)
    rs = np.random.RandomState(1000 + L * 7 + 3 + flsph + iwave * 100 + igr * 1000)
    nguard = 0
    for _ in range(n // 50000):
        nlay, h, vp, vs, rho = synth_models(rs, 50000, L, lvz_frac=0.25, ragged=True)
        v, e = lean.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr, flsph=flsph)
        assert lean.last_swd_kernel() == "lean"
        nguard += sum(lean.guard_stats()[0])
        ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr, flsph=flsph)
        check_against_the_reference(v, e, ov, oe)
    if iwave == 1:
        assert nguard > 0      # (the sets do contain the situations the guard is there for)


@pytest.mark.parametrize("ref", sorted(REFS))
def test_the_lane_per_evaluation_kernel_with_the_fast_arithmetic(lean, oracle, ref):
    """A call kept off the trial-per-lane kernel (here by the experiment switch that bounds its calls; also: forced lanes per
    model) takes swd_kernel's builds with the fast arithmetic -- the same guarantees."""
    rs = np.random.RandomState(4711)
    nlay, h, vp, vs, rho = synth_models(rs, 70000, 10, lvz_frac=0.25, ragged=True)
    per = np.linspace(2, 60, 30)
    iwave, igr = REFS[ref]
    lean.set_tuning("swd_lean_pairs", 65536)
    try:
        v, e = lean.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)
        assert lean.last_swd_kernel() == "lane"
    finally:
        lean.set_tuning("swd_lean_pairs", 0)
    ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr)
    check_against_the_reference(v, e, ov, oe)
    v4, e4 = lean.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)     # (the same call on the trial-per-lane kernel: 4 trials per round)
    assert lean.last_swd_kernel() == "lean"
    check_against_the_reference(v4, e4, ov, oe)


@pytest.mark.parametrize("trials", [4, 8, 16, 32, 64])
def test_every_trial_count_keeps_the_guarantees(lean, oracle, trials):
    """bh_engine_set_swd_trials: 4 / 8 trials per round, 32 / 64 (two models or one per wavefront) -- the same flags, zero rows and tolerance; the setting is validated and restored."""
    from bayhunter_amd.engine import EngineError
    rs = np.random.RandomState(99 + trials)
    nlay, h, vp, vs, rho = synth_models(rs, 3000, 12, lvz_frac=0.3, ragged=True)
    per = np.linspace(2, 60, 30)
    assert lean.swd_trials() == 0
    with pytest.raises(EngineError):
        lean.set_swd_trials(12)
    lean.set_swd_trials(trials)
    try:
        for iwave in (2, 1):
            v, e = lean.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0)
            assert lean.last_swd_kernel() == "lean"
            ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, 0)
            check_against_the_reference(v, e, ov, oe)
    finally:
        lean.set_swd_trials(0)


def test_reference_golden_vectors_within_tolerance(lean):
    g = golden("swd_golden.npz")
    nlay = g["nlay"]
    for pset in ("p21", "p30"):
        per = g["x_" + pset]
        for ir, ref in enumerate(g["refs"]):
            if str(ref) not in REFS:
                continue
            iwave, igr = REFS[str(ref)]
            vel, err = lean.swd_batch(nlay, g["h"], g["vp"], g["vs"], g["rho"], per, iwave, igr, layout="model_major")
            assert lean.last_swd_kernel() == "lean"
            ok = g["ok_" + pset][:, ir].astype(bool)
            assert np.array_equal(err == 0, ok)
            assert worst_rel(vel, g["y_" + pset][:, ir], ok) <= ACHIEVED


def test_a_model_alone_a_window_and_a_batch_give_the_same_bits(lean):
    """The result is a function of the model and of the trials per round (64 up to 1024 (model, target) pairs in a call; a sampler
    pins the number: DeviceChains): a window, a model evaluated alone and a permuted batch agree bit for bit."""
    rs = np.random.RandomState(77)
    nlay, h, vp, vs, rho = synth_models(rs, 1016, 12, lvz_frac=0.3, ragged=True)
    per = np.linspace(1.5, 70, 35)
    for iwave in (2, 1):
        va, ea = lean.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0)
        for b in (0, 17, 199, 1015):
            vb, eb = lean.swd_batch(nlay[b:b + 1], h[:, b:b + 1], vp[:, b:b + 1], vs[:, b:b + 1], rho[:, b:b + 1], per, iwave, 0)
            assert np.array_equal(vb[0], va[b]) and eb[0] == ea[b]
        sl = slice(100, 227)
        vw, ew = lean.swd_batch(nlay[sl], h[:, sl], vp[:, sl], vs[:, sl], rho[:, sl], per, iwave, 0)
        assert np.array_equal(vw, va[sl]) and np.array_equal(ew, ea[sl])
        perm = rs.permutation(1016)
        vp_, ep_ = lean.swd_batch(nlay[perm], h[:, perm], vp[:, perm], vs[:, perm], rho[:, perm], per, iwave, 0)
        assert np.array_equal(vp_, va[perm]) and np.array_equal(ep_, ea[perm])


@pytest.mark.parametrize("trials", [16, 32])
def test_result_is_a_function_of_the_model_and_the_trials(lean, trials):
    """What include/bh_engine.h promises of the default path: with the trials per round PINNED
    (bh_engine_set_swd_trials) a model's velocities and flag do not depend on the call it is in -- calls on either side of the
    shape thresholds (1024 / 5120 / 10240 (model, target) pairs), a model alone, a slice, prior-like models.  With the number
    left to the call's shape (0) the same model may differ in its last digits between shapes: asserted to stay within the
    reference's own 1e-6 scatter (2e-6), never a different flag."""
    from bayhunter_amd.synth import prior_models
    rs = np.random.RandomState(4242 + trials)
    a = synth_models(rs, 6000, 12, lvz_frac=0.3, ragged=True)
    b = prior_models(rs, 6000, 12)
    nlay, h, vp, vs, rho = [np.concatenate((x, y), axis=-1) for x, y in zip(a, b)]   # 12 000 models: 16 trials per round by shape
    per = np.linspace(2, 60, 30)
    cuts = [(0, 12000), (0, 900), (900, 5000), (5000, 5001), (6000, 6700), (6700, 12000)]   # 64 / 32 / 64 / 64 / 16 ... by shape
    for iwave in (2, 1):
        lean.set_swd_trials(trials)
        try:
            va, ea = lean.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0)
            assert lean.last_swd_kernel() == "lean"
            for lo, hi in cuts[1:]:
                sl = slice(lo, hi)
                v, e = lean.swd_batch(nlay[sl], h[:, sl], vp[:, sl], vs[:, sl], rho[:, sl], per, iwave, 0)
                assert np.array_equal(v, va[sl]) and np.array_equal(e, ea[sl]), (iwave, lo, hi)
        finally:
            lean.set_swd_trials(0)
        for lo, hi in cuts[1:]:       # by the call's shape: last digits only
            sl = slice(lo, hi)
            v, e = lean.swd_batch(nlay[sl], h[:, sl], vp[:, sl], vs[:, sl], rho[:, sl], per, iwave, 0)
            assert np.array_equal(e, ea[sl]) and np.array_equal(v == 0, va[sl] == 0)
            assert worst_rel(v, va[sl], (v != 0) & (va[sl] != 0)) <= ACHIEVED


def test_the_order_per_xcd_is_scheduling_only(lean, oracle):
    """Where the shapes divide (4096 models with 16 trials per round, 2048 with 32 ...) the models are ordered inside eight blocks of
    the batch, a block per XCD, and the second target runs them in the opposite order inside the XCDs (bh_tuning.h: swd_lean_xcd,
    swd_lean_flip): the same bits as with one order for the chip and as in batch order -- one target, two targets, a shape that
    does not divide -- and the reference's flags."""
    from bayhunter_amd import engine as E
    per = np.linspace(2, 60, 30)
    yobs = 3.4 + 0.01 * per
    rs = np.random.RandomState(123)
    try:
        for B in (4096, 2048, 4100, 512):
            nlay, h, vp, vs, rho = synth_models(rs, B, 10, lvz_frac=0.2)
            noise = np.tile([0, 0.05, 0, 0.05], (B, 1))
            out = {}
            for xcd, flip, nosort in ((1, 256, 0), (0, 256, 0), (1, 0, 0), (0, 0, 1)):
                lean.set_tuning("swd_lean_xcd", xcd)
                lean.set_tuning("swd_lean_flip", flip)
                lean.set_tuning("swd_lean_no_sort", nosort)
                lean.set_targets([dict(kind=E.TARGET_SWD, law=0, n=30, x=per, yobs=yobs, iwave=2, igr=0),
                                  dict(kind=E.TARGET_SWD, law=0, n=30, x=per, yobs=yobs, iwave=1, igr=0)])
                two = lean.evaluate_batch(nlay, h, vp, vs, noise, want_ymod=True)
                assert lean.last_swd_kernel() == "lean"
                one = lean.swd_batch(nlay, h, vp, vs, rho, per, 2, 0)
                out[(xcd, flip, nosort)] = (two, one)
            ref = out[(0, 0, 1)]
            for k, (two, one) in out.items():
                for a, b in zip(two, ref[0]):
                    assert np.array_equal(a, b, equal_nan=True), (B, k)
                assert np.array_equal(one[0], ref[1][0]) and np.array_equal(one[1], ref[1][1]), (B, k)
            ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, 2, 0)
            check_against_the_reference(ref[1][0], ref[1][1], ov, oe)
    finally:
        lean.set_tuning("swd_lean_xcd", 1)
        lean.set_tuning("swd_lean_flip", 256)
        lean.set_tuning("swd_lean_no_sort", 0)


def test_earth_flattening_deep_models_and_many_periods(lean, oracle):
    rs = np.random.RandomState(9)
    for L, B, fl, K in ((12, 300, 1, 30), (32, 60, 0, 60), (21, 100, 0, 7), (50, 40, 0, 30)):
        per = np.linspace(2, 60, K)
        nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=0.2, ragged=(L <= 21), hmin=0.5 if L > 12 else 1.5, hmax=2.0 if L > 12 else 8.0)
        for iwave in (2, 1):
            v, e = lean.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0, flsph=fl)
            assert (lean.last_swd_kernel() == "lean") == (L <= 32)     # (deeper arrays: the other kernels' builds)
            ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, 0, flsph=fl)
            check_against_the_reference(v, e, ov, oe)


def test_broken_models_and_water_layers(lean, oracle):
    """NaN / infinite / negative / absurd parameters are reported as failed without being searched, the healthy models of
    the batch are unaffected; a model with a water layer (vs = 0 on top) takes the reference's sequence."""
    rs = np.random.RandomState(8)
    nlay, h, vp, vs, rho = synth_models(rs, 24, 6)
    vs[2, 0] = np.nan; vp[1, 1] = np.inf; h[0, 2] = np.nan; rho[3, 3] = -1.0; vs[:, 4] = -1.0; h[1, 5] = -5.0
    vp[:, 6] = 1e-60; vs[:, 7] = 1e30
    vs[0, 8] = 0.0; vp[0, 8] = 1.5; rho[0, 8] = 1.03     # water on top of model 8
    per = np.linspace(2, 40, 12)
    good = np.arange(8, 24)
    for iwave in (2, 1):
        v, e = lean.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0)
        assert lean.last_swd_kernel() == "lean"
        assert (e[:8] == 1).all() and (v[:8] == 0).all()
        ov, oe, _ = oracle.swd_batch(nlay[good], h.T[good], vp.T[good], vs.T[good], rho.T[good], per, iwave, 0)
        check_against_the_reference(v[good], e[good], ov, oe)
        assert np.array_equal(v[8], ov[0])                  # (the water-layer model: the reference's bits)
        assert lean.guard_stats()[0][0] >= 1


def test_empty_and_single_period_calls(lean, oracle):
    rs = np.random.RandomState(5)
    nlay, h, vp, vs, rho = synth_models(rs, 9, 7)
    v, e = lean.swd_batch(nlay, h, vp, vs, rho, np.array([12.5]), 2, 0)
    ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, np.array([12.5]), 2, 0)
    check_against_the_reference(v, e, ov, oe)
    v0, e0 = lean.swd_batch(nlay[:0], h[:, :0], vp[:, :0], vs[:, :0], rho[:, :0], np.array([12.5]), 2, 0)
    assert v0.shape[0] == 0 and e0.shape[0] == 0
