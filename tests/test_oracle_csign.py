"""The certified-sign scan on the CPU (oracle/csign_oracle.c, oracle/swd_oracle.c: bracket_and_refine with g_prescan): NOT the
reference's algorithm but the engine's -- restated so that (1) its error bound can be held against the reference-exact
recursion, (2) the claim "same brackets, same bits" can be checked against the step-by-step scan the oracle is pinned with
(tests/test_oracle_swd.py: bit-equal to the compiled reference), for every target type, higher modes, earth flattening
and both root refinements."""
import numpy as np
import pytest

from bayhunter_amd.synth import synth_models

PER = np.linspace(2, 60, 30)


def _models(seed, n, L, **kw):
    rs = np.random.RandomState(seed)
    nlay, h, vp, vs, rho = synth_models(rs, n, L, **kw)
    return rs, nlay, h, vp, vs, rho


@pytest.mark.parametrize("ifunc", [1, 2])
def test_error_bound_holds_against_the_reference_exact_recursion(oracle, ifunc):
    """Random models of 2..14 layers, trial velocities from below the slowest to above the fastest S velocity -- among them
    velocities within 1e-3 ... 1e-9 relative of a layer velocity, where the reference's own k - k_beta cancels: the surface
    vector of the cheap evaluation lies within its bound of the reference-exact one (after the common positive scale), every
    certified sign is the reference-exact sign, and nearly every point is certified."""
    rs, nlay, h, vp, vs, rho = _models(17 + ifunc, 60, 14, lvz_frac=0.3, ragged=True)
    worst, ncert, ntot, wrong = 0.0, 0, 0, 0
    for b in range(nlay.size):
        L = int(nlay[b])
        d, a, bb, r = [x[:L, b].astype(np.float32) for x in (h, vp, vs, rho)]
        cs = list(rs.uniform(0.8 * bb.min(), 1.02 * bb.max(), 40))
        for v in np.concatenate([bb, a]):
            for rel in (1e-3, 1e-6, 1e-9):
                cs.append(float(v) * (1 + rel * rs.choice([-1, 1])))
        for T in rs.choice(PER, 3, replace=False):
            om = 2 * np.pi / T
            for c in cs:
                ok, ev, ep = oracle.csign(ifunc, om, c, d, a, bb, r, vec=True)
                if not (np.all(np.isfinite(ev)) and np.all(np.isfinite(ep))):
                    assert not ok
                    continue
                v64 = oracle.secular_vec(ifunc, om, c, d, a, bb, r)
                im = int(np.argmax(np.abs(ev)))
                s = ev[im] / v64[im]                      # the positive scale between the two normalisations
                assert s > 0
                allow = ep + ep[im] * np.abs(ev)          # (+ the scale's own uncertainty)
                worst = max(worst, float(np.max(np.abs(ev - s * v64) / np.maximum(allow, 1e-300))))
                ntot += 1
                if ok:
                    ncert += 1
                    wrong += (ev[0] < 0) != (v64[0] < 0)
    assert wrong == 0
    assert worst <= 0.5, worst                            # (the final test carries another factor 2)
    assert ncert >= 0.995 * ntot, (ncert, ntot)


@pytest.mark.parametrize("fast", [0, 2])
def test_certified_scan_returns_the_bits_of_the_step_by_step_scan(oracle, fast):
    """bho_swd_batch with and without the certified-sign scan: identical velocities and failure flags for all four target
    types, higher modes, earth flattening, ragged LVZ-rich models, models that fail and models with a water layer (never
    certified); the landings replace most of the scan's reference-exact evaluations, and none of them is contradicted by
    the reference-exact values at the landing points."""
    rs, nlay, h, vp, vs, rho = _models(5, 160, 12, lvz_frac=0.3, ragged=True)
    vs[0, :6] = 0.0
    vp[0, :6] = 1.5
    vs[:, 6:10] *= 0.2
    a = [np.ascontiguousarray(x.T) for x in (h, vp, vs, rho)]
    per = np.sort(rs.uniform(1.0, 80.0, 24))
    with oracle.swd_search(fast):
        for iwave in (1, 2):
            for igr, mode, flsph in ((0, 1, 0), (1, 1, 0), (0, 3, 0), (1, 2, 1)):
                v0, e0, n0 = oracle.swd_batch(nlay, *a, per, iwave, igr, mode=mode, flsph=flsph, nthreads=1)
                oracle.swd_prescan_stats()
                with oracle.swd_prescan(1):
                    v1, e1, n1 = oracle.swd_batch(nlay, *a, per, iwave, igr, mode=mode, flsph=flsph, nthreads=1)
                st = oracle.swd_prescan_stats()
                assert np.array_equal(v0, v1) and np.array_equal(e0, e1), (fast, iwave, igr, mode, flsph)
                assert st[3] == 0                          # no landing contradicted
                assert st[2] > 0 and n1 < 0.8 * n0, (st, n0, n1)


def test_certified_scan_on_the_bench_models(oracle):
    """BASELINE configs[1]'s models (10 layers, 30 periods): one landing per period, the reference-exact evaluations of the
    default search fall from ~19 to ~5 per period."""
    rs, nlay, h, vp, vs, rho = _models(20260927, 64, 10, lvz_frac=0.1)
    a = [np.ascontiguousarray(x.T) for x in (h, vp, vs, rho)]
    for iwave in (2, 1):
        with oracle.swd_search(2):
            v0, e0, n0 = oracle.swd_batch(nlay, *a, PER, iwave, 0, nthreads=1)
            oracle.swd_prescan_stats()
            with oracle.swd_prescan(1):
                v1, e1, n1 = oracle.swd_batch(nlay, *a, PER, iwave, 0, nthreads=1)
            st = oracle.swd_prescan_stats()
        assert np.array_equal(v0, v1) and np.array_equal(e0, e1)
        assert st[2] >= 0.95 * 30 * nlay.size and n1 <= 0.35 * n0, (st, n0, n1)
