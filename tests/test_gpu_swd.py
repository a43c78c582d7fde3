"""Parity of the HIP dispersion path (bh_swd_batch through the C ABI) with the oracle, the
reference golden vectors and the reference's tutorial files.  Tolerance of north_star: 1e-5
relative on dispersion velocities (stated below); in practice the results are bit-identical."""
import numpy as np
import pytest

from conftest import golden, st3
from bayhunter_amd.synth import synth_models

pytestmark = pytest.mark.gpu
REFS = {"rdispph": (2, 0), "rdispgr": (2, 1), "ldispph": (1, 0), "ldispgr": (1, 1)}
RTOL = 1e-5
# Model 73 of the golden set (vs = 3.9/1.6/4.4/3.0 km/s: a 1.6 km/s channel over a slow
# half-space) is search-chaotic: surf96 jumps between modes from period to period and its
# Neville refinement is ill-conditioned there.  With the device library's sin/cos/exp (1 ulp from
# glibc) the number of refinement steps changed (812 vs 813 evaluations) and the group velocity
# moved by 1.1e-4 at one period.  Since the dispersion kernels use the glibc-exact restatement
# (csrc/bh_libm.h) this model, like every other, is reproduced bit for bit; it stays in the
# tests as the canary for that property.


def compare(vel, err, ovel, oerr):
    assert np.array_equal(err, oerr)
    ok = oerr == 0
    rel = np.abs(vel[ok] - ovel[ok]) / np.abs(ovel[ok])
    assert rel.size == 0 or rel.max() <= RTOL, rel.max()
    # failed models: zeros from the failing period on, like surfdisp96.f:348-354
    assert np.array_equal(vel[~ok] == 0, ovel[~ok] == 0)
    good = (ovel[~ok] != 0)
    if good.any():
        assert np.max(np.abs(vel[~ok][good] - ovel[~ok][good]) / np.abs(ovel[~ok][good])) <= RTOL
    return rel


@pytest.mark.parametrize("ref", sorted(REFS))
def test_random_ragged_models_match_oracle(engine, oracle, ref):
    rs = np.random.RandomState(101)
    nlay, h, vp, vs, rho = synth_models(rs, 777, 21, lvz_frac=0.25, ragged=True)
    per = np.linspace(2, 60, 30)
    iwave, igr = REFS[ref]
    vel, err = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)
    ovel, oerr, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr)
    rel = compare(vel, err, ovel, oerr)
    assert np.all(rel == 0)  # bit-identical


@pytest.mark.parametrize("pset", ["p21", "p30"])
def test_reference_golden_vectors(engine, pset):
    g = golden("swd_golden.npz")
    per = g["x_" + pset]
    nlay = g["nlay"]
    for ir, ref in enumerate(g["refs"]):
        iwave, igr = REFS[str(ref)]
        vel, err = engine.swd_batch(nlay, g["h"], g["vp"], g["vs"], g["rho"], per, iwave, igr, layout="model_major")
        ok = g["ok_" + pset][:, ir].astype(bool)
        assert np.array_equal(err == 0, ok)
        ref_y = g["y_" + pset][:, ir]
        assert np.max(np.abs(vel[ok] - ref_y[ok]) / np.abs(ref_y[ok])) <= RTOL   # north_star bar
        assert np.array_equal(vel[ok], ref_y[ok])                                # what is achieved


@pytest.mark.parametrize("ref", sorted(REFS))
def test_tutorial_files(engine, ref):
    x, y = st3(ref)
    h = np.array([[5., 23., 8., 0.]]).T; vs = np.array([[2.7, 3.6, 3.8, 4.4]]).T; vp = vs * 1.73
    vel, err = engine.swd_batch(np.array([4]), h, vp, vs, vp * 0.32 + 0.77, x, *REFS[ref])
    assert err[0] == 0 and np.max(np.abs(vel[0] - y)) <= 5.1e-5


@pytest.mark.parametrize("B", [1, 63, 64, 65, 130])
def test_batch_sizes_and_layouts(engine, oracle, B):
    rs = np.random.RandomState(B)
    nlay, h, vp, vs, rho = synth_models(rs, B, 7, ragged=True)
    per = np.linspace(3, 40, 11)
    v1, e1 = engine.swd_batch(nlay, h, vp, vs, rho, per, 2, 0)
    v2, e2 = engine.swd_batch(nlay, h.T.copy(), vp.T.copy(), vs.T.copy(), rho.T.copy(), per, 2, 0, layout="model_major")
    assert np.array_equal(v1, v2) and np.array_equal(e1, e2)
    ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, 2, 0)
    compare(v1, e1, ov, oe)


def test_edge_cases(engine, oracle):
    rs = np.random.RandomState(3)
    # empty batch, single period, 60 periods (the Fortran's NP), two-layer models, padded Lmax
    nlay, h, vp, vs, rho = synth_models(rs, 5, 4)
    v, e = engine.swd_batch(nlay[:0], h[:, :0], vp[:, :0], vs[:, :0], rho[:, :0], np.array([10.]), 2, 0)
    assert v.shape == (0, 1) and e.shape == (0,)
    for per in (np.array([7.5]), np.linspace(1, 80, 60)):
        for iwave, igr in REFS.values():
            v, e = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)
            ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr)
            compare(v, e, ov, oe)
    nl2, h2, vp2, vs2, rho2 = synth_models(rs, 9, 2)
    pad = lambda a: np.vstack([a, np.full((98, a.shape[1]), 123.0)])  # Lmax = 100, garbage beyond nlay
    v, e = engine.swd_batch(nl2, pad(h2), pad(vp2), pad(vs2), pad(rho2), np.linspace(2, 30, 8), 1, 1)
    ov, oe, _ = oracle.swd_batch(nl2, h2.T, vp2.T, vs2.T, rho2.T, np.linspace(2, 30, 8), 1, 1)
    compare(v, e, ov, oe)


def test_failing_models_are_reported_in_band(engine, oracle):
    """Strong velocity inversion: surf96 finds no root -> err=1 and zeros, never an exception."""
    h = np.array([[2., 3., 10., 0.]]).T; vs = np.array([[3.9, 1.6, 4.4, 3.0]]).T; vp = vs * 1.75
    rho = vp * 0.32 + 0.77
    per = np.linspace(2, 60, 30)
    for iwave, igr in REFS.values():
        v, e = engine.swd_batch(np.array([4]), h, vp, vs, rho, per, iwave, igr)
        ov, oe, _ = oracle.swd_batch(np.array([4]), h.T, vp.T, vs.T, rho.T, per, iwave, igr)
        assert np.array_equal(e, oe) and np.array_equal(v, ov)  # golden model 73: bit for bit
    assert e[0] == 1  # Love finds no root at all beyond the 7th period


def test_parity_statistics_lvz_rich(engine, oracle):
    """20k models, a quarter with a low-velocity layer, all four wave/velocity types."""
    rs = np.random.RandomState(2024)
    nlay, h, vp, vs, rho = synth_models(rs, 20000, 12, lvz_frac=0.25, ragged=True)
    per = np.linspace(2, 60, 30)
    for (iwave, igr) in REFS.values():
        v, e = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)
        ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr)
        assert np.array_equal(e, oe)
        ok = oe == 0
        rel = np.abs(v[ok] - ov[ok]) / ov[ok]
        assert np.array_equal(v[ok], ov[ok])  # bit-identical, every model, every period


def test_bad_arguments_fail_loudly(engine):
    from bayhunter_amd.engine import EngineError
    nlay, h, vp, vs, rho = synth_models(np.random.RandomState(1), 2, 3)
    with pytest.raises(EngineError):
        engine.swd_batch(nlay, h, vp, vs, rho, np.linspace(1, 100, 61), 2, 0)  # > 60 periods (NP = 60)
    with pytest.raises(EngineError):
        engine.swd_batch(nlay, h, vp, vs, rho, np.linspace(1, 30, 5), 3, 0)    # iwave must be 1 or 2


@pytest.mark.parametrize("mode", [2, 3])
def test_higher_modes_match_oracle_and_reference(engine, oracle, mode):
    """mode > 1 (surfdisp96.f:219-357): all modes up to `mode` are searched in turn, the last one
    is returned, missing higher-mode roots leave zeros without setting err."""
    rs = np.random.RandomState(40 + mode)
    nlay, h, vp, vs, rho = synth_models(rs, 200, 12, lvz_frac=0.2, ragged=True)
    per = np.linspace(2, 60, 30)
    for iwave, igr in REFS.values():
        v, e = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr, mode=mode)
        ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr, mode=mode)
        assert np.array_equal(e, oe) and np.array_equal(v, ov)
    if mode == 2:
        g = golden("swd_golden.npz")
        sub = g["sub_idx"]
        for ir, ref in enumerate(g["refs"]):
            iwave, igr = REFS[str(ref)]
            v, e = engine.swd_batch(g["nlay"][sub], g["h"][sub], g["vp"][sub], g["vs"][sub], g["rho"][sub], g["x_p30"],
                                    iwave, igr, mode=2, layout="model_major")
            ok = g["ok_mode2"][:, ir].astype(bool)
            assert np.array_equal(e == 0, ok)
            assert np.array_equal(v[ok], g["y_mode2"][:, ir][ok])


def test_earth_flattening_matches_oracle_and_reference(engine, oracle):
    """flsph = 1 (surfdisp96.f:486-553).  The transform's log and powf are restatements of the host
    libm's (csrc/bh_libm.h), so this option is bit-identical too."""
    rs = np.random.RandomState(50)
    nlay, h, vp, vs, rho = synth_models(rs, 2000, 12, ragged=True, lvz_frac=0.2)
    per = np.linspace(2, 60, 30)
    for iwave, igr in REFS.values():
        v, e = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr, flsph=1)
        ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr, flsph=1)
        assert np.array_equal(e, oe)
        assert np.array_equal(v, ov), (iwave, igr)
    g = golden("swd_golden.npz")
    sub = g["sub_idx"]
    for ir, ref in enumerate(g["refs"]):
        iwave, igr = REFS[str(ref)]
        v, e = engine.swd_batch(g["nlay"][sub], g["h"][sub], g["vp"][sub], g["vs"][sub], g["rho"][sub], g["x_p30"],
                                iwave, igr, flsph=1, layout="model_major")
        ok = g["ok_sph"][:, ir].astype(bool)
        assert np.array_equal(e == 0, ok)
        assert np.array_equal(v[ok], g["y_sph"][:, ir][ok])


def test_device_libm_restatement_is_bit_identical_to_host_libm(engine, oracle):
    """csrc/bh_libm.h (what the dispersion kernels call) against the host's libm sincos()/exp(),
    which is what the reference's compiled Fortran calls: identical bits, all argument ranges."""
    rs = np.random.RandomState(3)
    for lo, hi in ((0, 1e-8), (0, 0.126), (0.1, 0.9), (0.8, 2.5), (2.4, 10), (0, 40), (0, 1e3), (1e3, 1e5), (1e5, 1.05e8)):
        x = rs.uniform(lo, hi, 400000) * rs.choice([-1, 1], 400000)
        assert np.array_equal(engine.probe_math(8, x).view(np.int64), oracle.libm_probe(0, x).view(np.int64))
        assert np.array_equal(engine.probe_math(9, x).view(np.int64), oracle.libm_probe(1, x).view(np.int64))
    for lo, hi in ((-1e-10, 0), (-1, 0), (-40, 0), (-130, 0), (-500, 500)):
        x = rs.uniform(lo, hi, 400000)
        assert np.array_equal(engine.probe_math(10, x).view(np.int64), oracle.libm_probe(2, x).view(np.int64))


def test_device_math_is_close_to_host_libm(engine):
    """Documents SURVEY.md 7 'device libm' (ocml, used outside the dispersion kernels): sqrt and 1/x
    are correctly rounded, sin/cos/exp within 1 ulp of glibc."""
    rs = np.random.RandomState(0)
    x = rs.uniform(0.01, 40, 50000)
    assert np.array_equal(engine.probe_math(0, x), np.sqrt(x))
    assert np.array_equal(engine.probe_math(5, x), 1.0 / x)
    for op, f, arg in ((1, np.sin, x), (2, np.cos, x), (3, np.exp, -x)):
        g, r = engine.probe_math(op, arg), f(arg)
        assert np.max(np.abs(g - r) / np.spacing(np.abs(r))) <= 1.0


def test_shared_reciprocal_division_is_exact(engine):
    """bh_quot(a, b, bh_rcp_refined(b)) == a / b bit for bit over the range the kernels use it in
    ([2^-400, 2^400], see csrc/bh_device.h), and the device's plain division is IEEE."""
    rs = np.random.RandomState(77)
    for rep in range(8):
        n = 1 << 18
        ea = rs.uniform(-400, 400, n); eb = rs.uniform(-400, 400, n)
        if rep % 4 == 1:
            ea = eb + rs.uniform(-60, 0.0, n)            # |a| <= |b|: the max-norm rescale
        if rep % 4 == 2:
            ea = np.clip(eb + rs.uniform(-3, 3, n), -400, 400)
        a = np.ldexp(rs.uniform(1, 2, n), ea.astype(int)) * rs.choice([-1, 1], n)
        b = np.ldexp(rs.uniform(1, 2, n), eb.astype(int)) * rs.choice([-1, 1], n)
        if rep % 4 == 3:                                  # mantissas next to 1 and 2
            a = np.ldexp(1 + rs.randint(0, 8, n) * 2.0 ** -52, ea.astype(int))
            b = np.ldexp(2 - rs.randint(1, 8, n) * 2.0 ** -52, eb.astype(int))
        pairs = np.column_stack((a, b)).ravel()
        fast, plain = engine.probe_math(6, pairs), engine.probe_math(7, pairs)
        assert np.array_equal(plain, a / b)
        assert np.array_equal(fast.view(np.int64), plain.view(np.int64))


@pytest.mark.parametrize("G", [1, 2, 5, 9, 16, 21])
def test_lane_mappings_return_identical_bits(engine, G):
    """One lane per model, or G lanes per model (any G): same velocities, same error flags."""
    rs = np.random.RandomState(31)
    nlay, h, vp, vs, rho = synth_models(rs, 300, 14, lvz_frac=0.3, ragged=True)
    per = np.linspace(2, 60, 30)
    try:
        for iwave, igr in REFS.values():
            engine.set_swd_group(1)
            v1, e1 = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)
            engine.set_swd_group(G)
            v2, e2 = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)
            assert np.array_equal(v1, v2) and np.array_equal(e1, e2)
    finally:
        engine.set_swd_group(0)


@pytest.mark.parametrize("G,J", [(5, 2), (9, 3), (9, 7), (13, 4), (16, 2), (21, 3), (1, 2), (1, 4), (1, 8), (1, 16)])
def test_lookahead_does_not_change_results(engine, G, J):
    """Look-ahead (extra lane groups -- with G = 1: extra lanes -- evaluating the trial velocities the root search will probably
    ask for next, bh_engine_set_swd_lookahead) only changes WHEN a secular value is computed, never
    which values the search consumes: velocities (all wave/velocity types, higher modes, failing
    models, ragged layer counts, water layers) and the number of consumed evaluations are unchanged."""
    rs = np.random.RandomState(77)
    nlay, h, vp, vs, rho = synth_models(rs, 200, 12, lvz_frac=0.3, ragged=True)
    vs[0, :8] = 0.0                      # a few models with a water layer on top
    vp[0, :8] = 1.5
    vs[:, 8:12] *= 0.2                   # and some that fail the search
    per = np.linspace(1.5, 70, 35)
    try:
        for iwave, igr in REFS.values():
            for mode in (1, 3):
                engine.set_swd_group(G)
                engine.set_swd_lookahead(1)
                engine.set_instrumentation(False, True)
                v1, e1 = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr, mode=mode)
                n1 = engine.last_neval()
                engine.set_swd_lookahead(J)
                v2, e2 = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr, mode=mode)
                n2 = engine.last_neval()
                assert np.array_equal(v1, v2) and np.array_equal(e1, e2), (iwave, igr, mode)
                assert n1 == n2 and n1 > 0
    finally:
        engine.set_swd_group(0)
        engine.set_swd_lookahead(0)
        engine.set_instrumentation(False, False)


@pytest.mark.parametrize("L,B", [(100, 5), (21, 300), (50, 40)])
def test_deep_models_up_to_the_fortran_limit(engine, oracle, L, B):
    """Models with up to NL = 100 layers (surfdisp96.f:59) -- the lanes-per-model choice has to adapt to
    what fits the LDS -- and the 20-layer transdimensional case of BASELINE configs[4]; all types."""
    rs = np.random.RandomState(L)
    nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=0.2, ragged=(L == 21))
    h[:-1] *= 10.0 / L                       # keep the stack ~60 km thick
    per = np.linspace(2, 60, 24)
    for iwave, igr in REFS.values():
        v, e = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)
        ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr)
        assert np.array_equal(e, oe) and np.array_equal(v, ov), (L, iwave, igr)


def test_broken_models_are_failed_in_band(engine, oracle):
    """NaN / infinite / negative / absurd parameters (a caller's bug).  The reference bounds its loops only
    through the model's velocities and walks the velocity axis for ever on such input; the kernels report
    such a model as failed (err = 1, zeros) without searching, and the healthy models of the batch are
    unaffected -- for every lane mapping and look-ahead setting the planner can choose."""
    rs = np.random.RandomState(8)
    nlay, h, vp, vs, rho = synth_models(rs, 24, 6)
    vs[2, 0] = np.nan; vp[1, 1] = np.inf; h[0, 2] = np.nan; rho[3, 3] = -1.0; vs[:, 4] = -1.0; h[1, 5] = -5.0
    vp[:, 6] = 1e-60; vs[:, 7] = 1e30
    per = np.linspace(2, 40, 12)
    good = np.arange(8, 24)
    try:
        for G, J in ((0, 0), (1, 1), (1, 4), (9, 1), (9, 2), (9, 7), (16, 4)):
            engine.set_swd_group(G)
            engine.set_swd_lookahead(J)
            for iwave, igr in REFS.values():
                v, e = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)
                assert (e[:8] == 1).all() and (v[:8] == 0).all()
                ov, oe, _ = oracle.swd_batch(nlay[good], h.T[good], vp.T[good], vs.T[good], rho.T[good], per, iwave, igr)
                assert np.array_equal(v[good], ov) and np.array_equal(e[good], oe)
    finally:
        engine.set_swd_group(0)
        engine.set_swd_lookahead(0)


def test_experiment_switches_are_a_table_with_an_api(engine):
    """csrc/bh_tuning.h: the library's experiment switches are parsed once per process; bh_engine_set_tuning changes one by name,
    unknown names are refused, and none of them changes a result (here: the progress board and the SIMD-pairing order)."""
    from bayhunter_amd.engine import EngineError
    assert engine.tuning("rf_no_cut") == 0 and engine.tuning("swd_rerun_wgs") == 256
    with pytest.raises(EngineError):
        engine.set_tuning("no_such_switch", 1)
    rs = np.random.RandomState(3)
    nlay, h, vp, vs, rho = synth_models(rs, 4096, 10, lvz_frac=0.1)
    per = np.linspace(2, 60, 30)
    v0, e0 = engine.swd_batch(nlay, h, vp, vs, rho, per, 2, 0)
    for name in ("swd_no_board", "swd_no_fair", "swd_no_simple"):
        engine.set_tuning(name, 1)
        try:
            assert engine.tuning(name) == 1
            v1, e1 = engine.swd_batch(nlay, h, vp, vs, rho, per, 2, 0)
        finally:
            engine.set_tuning(name, 0)
        assert np.array_equal(v0, v1) and np.array_equal(e0, e1), name


@pytest.mark.parametrize("iwave", [1, 2])
def test_group_velocity_chains_as_two_launches_return_the_same_bits(engine, oracle, iwave):
    """Fundamental-mode group velocities run as the chain of the first roots (t/(1+h)), then one independent search per (model,
    period) for the second roots (t/(1-h), surfdisp96.f:282-287) -- bh_engine.hip launch_swd_jobs; bh_tuning.h swd_gsplit = 0
    keeps the single launch.  Both ways, every batch shape, flattened or not: the oracle's bits."""
    rs = np.random.RandomState(77 + iwave)
    assert engine.tuning("swd_gsplit") == 1 << 24
    try:
        for B, L, K, flsph in ((1, 6, 30, 0), (7, 12, 5, 1), (300, 21, 30, 0), (2600, 9, 21, 0), (9000, 10, 30, 0)):
            nlay, h, vp, vs, rho = synth_models(rs, B, L, lvz_frac=0.3, ragged=True)
            per = np.sort(rs.uniform(1.5, 70.0, K))
            ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, 1, flsph=flsph)
            for cap in (0, 1 << 24):
                engine.set_tuning("swd_gsplit", cap)
                v, e = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, 1, flsph=flsph)
                assert np.array_equal(e, oe), (B, cap)
                assert np.array_equal(v, ov), (B, cap)
    finally:
        engine.set_tuning("swd_gsplit", 1 << 24)
