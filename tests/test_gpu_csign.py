"""The certified-sign scan on the device (csrc/swd_csign.h, SearchT<.., PRE>, bh_engine_set_swd_prescan; opt-in).
(1) the device's certified-sign evaluation equals the oracle's restatement (oracle/csign_oracle.c) BIT FOR BIT -- value, bound
and the certified flag, i.e. the same fallback decisions; (2) with the scan on, velocities and failure flags are bit-identical
to the step-by-step scan for every target type, higher modes, earth flattening, both root refinements and every launch plan,
with far fewer reference-exact evaluations."""
import numpy as np
import pytest

from bayhunter_amd.synth import synth_models

pytestmark = pytest.mark.gpu
REFS = {"rdispph": (2, 0), "rdispgr": (2, 1), "ldispph": (1, 0), "ldispgr": (1, 1)}


@pytest.mark.parametrize("iwave", [1, 2])
def test_device_evaluation_equals_the_cpu_restatement_bit_for_bit(engine, oracle, iwave):
    rs = np.random.RandomState(41 + iwave)
    nlay, h, vp, vs, rho = synth_models(rs, 40, 21, lvz_frac=0.3, ragged=True)
    per = np.linspace(1.5, 70, 35)
    ntot = ncert = 0
    for b in range(nlay.size):
        L = int(nlay[b])
        d, a, bb, r = [x[:L, b].astype(np.float32) for x in (h, vp, vs, rho)]
        cs = list(rs.uniform(0.8 * bb.min(), 1.02 * bb.max(), 60))
        for v in np.concatenate([bb, a]):                 # and velocities right at a layer velocity
            for rel in (0.0, 1e-4, 1e-8):
                cs.append(float(v) * (1 + rel * rs.choice([-1, 1])))
        c = np.tile(np.array(cs), 3)
        om = np.repeat(2 * np.pi / rs.choice(per, 3, replace=False), len(cs))
        val, bd, ok = engine.probe_csign(iwave, d, a, bb, r, om, c)
        for i in range(c.size):
            ook, ov, ob = oracle.csign(iwave, om[i], c[i], d, a, bb, r)
            assert ok[i] == ook, (b, i)
            same = lambda x, y: (x == y) or (np.isnan(x) and np.isnan(y))
            assert same(val[i], ov) and same(bd[i], ob), (b, i, val[i], ov, bd[i], ob)
        ntot += c.size
        ncert += int(ok.sum())
    assert ncert > 0.75 * ntot   # (a tenth of the points sit right at a layer velocity: never certified)


@pytest.mark.parametrize("search", ["reference", "fast"])
def test_same_bits_with_and_without_the_certified_scan(engine, oracle, search):
    rs = np.random.RandomState(5)
    nlay, h, vp, vs, rho = synth_models(rs, 6000, 12, lvz_frac=0.3, ragged=True)   # (several models per wavefront: where the scan applies)
    vs[0, :8] = 0.0                      # a few models with a water layer on top (never certified)
    vp[0, :8] = 1.5
    vs[:, 8:12] *= 0.2                   # and some that fail the search
    per = np.sort(rs.uniform(1.0, 80.0, 30))
    a = [np.ascontiguousarray(x.T) for x in (h, vp, vs, rho)]
    assert not engine.swd_prescan()      # opt-in (measured, it does not pay at the BASELINE batch)
    engine.set_swd_search(search)
    engine.set_swd_scan("steps")         # (like with like: the step-by-step scan with and without the look-ahead)
    engine.set_instrumentation(False, True)
    try:
        for (iwave, igr) in REFS.values():
            for mode, flsph in ((1, 0), (3, 0), (2, 1)):
                got = {}
                for on in (True, False):
                    with engine.prescanning(on):
                        got[on] = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr, mode=mode, flsph=flsph) + (engine.last_neval(),)
                assert np.array_equal(got[True][0], got[False][0]) and np.array_equal(got[True][1], got[False][1]), (iwave, igr, mode, flsph)
                assert got[True][2] < 0.8 * got[False][2], (iwave, igr, mode, got[True][2], got[False][2])
                if search == "reference":   # and they are the reference's bits (the oracle's step-by-step scan)
                    ov, oe, _ = oracle.swd_batch(nlay, *a, per, iwave, igr, mode=mode, flsph=flsph)
                    assert np.array_equal(got[True][0], ov) and np.array_equal(got[True][1], oe)
    finally:
        engine.set_instrumentation(False, False)
        engine.set_swd_search("reference")
        engine.set_swd_scan("auto")


@pytest.mark.parametrize("G,J", [(5, 2), (9, 1), (9, 2), (9, 3), (9, 7), (16, 2), (21, 3), (0, 0)])
def test_certified_scan_does_not_depend_on_the_launch_plan(engine, G, J):
    """Lanes per model and trials per round decide how many grid points one look covers -- not what the search returns."""
    rs = np.random.RandomState(77)
    nlay, h, vp, vs, rho = synth_models(rs, 3000, 12, lvz_frac=0.3, ragged=True)
    vs[:, 8:12] *= 0.2
    per = np.linspace(1.5, 70, 35)
    try:
        for search in ("fast", "reference"):
            engine.set_swd_search(search)
            for iwave in (2, 1):
                engine.set_swd_group(9)
                engine.set_swd_lookahead(2)
                with engine.prescanning(False):
                    v1, e1 = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0)
                engine.set_swd_group(G)
                engine.set_swd_lookahead(J)
                with engine.prescanning(True):
                    v2, e2 = engine.swd_batch(nlay, h, vp, vs, rho, per, iwave, 0)
                assert np.array_equal(v1, v2) and np.array_equal(e1, e2), (search, iwave)
    finally:
        engine.set_swd_group(0)
        engine.set_swd_lookahead(0)
        engine.set_swd_search("reference")


def test_bench_batch_with_the_certified_scan_equals_the_plain_scan(engine):
    """BASELINE configs[1]'s batch (4096 ten-layer models, Rayleigh + Love in ONE fused launch, the default search) with the
    certified scan against the same call without: the same bits, a third of the reference-exact evaluations."""
    from bayhunter_amd import engine as E
    rs = np.random.RandomState(20260927)
    nlay, h, vp, vs, rho = synth_models(rs, 4096, 10, lvz_frac=0.1)
    per = np.linspace(2, 60, 30)
    yobs = 3.4 + 0.01 * per
    engine.set_targets([{"kind": E.TARGET_SWD, "law": E.LAW_NOCORR, "n": 30, "x": per, "yobs": yobs, "iwave": 2, "igr": 0},
                        {"kind": E.TARGET_SWD, "law": E.LAW_NOCORR, "n": 30, "x": per, "yobs": yobs, "iwave": 1, "igr": 0}])
    noise = np.tile(np.array([0.0, 0.05, 0.0, 0.05]), (4096, 1))
    engine.set_swd_search("fast")
    engine.set_instrumentation(False, True)
    try:
        got = {}
        for on in (True, False):
            with engine.prescanning(on):
                got[on] = engine.evaluate_batch(nlay, h, vp, vs, noise, want_ymod=True) + (engine.last_neval(),)
        for k in range(4):
            assert np.array_equal(got[True][k], got[False][k]), k
        assert got[True][4] < 0.4 * got[False][4], (got[True][4], got[False][4])
    finally:
        engine.set_instrumentation(False, False)
        engine.set_swd_search("reference")
