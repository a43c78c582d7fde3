"""Parity of the HIP receiver-function path (bh_rf_batch) with the oracle and the reference
golden vectors.  north_star tolerance: 1e-4 on RF amplitudes; asserted here: 1e-9 of the peak."""
import numpy as np
import pytest

from conftest import golden, st3
from bayhunter_amd.synth import synth_models

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.mark.parametrize("nsamp,fsamp,nkeep", [(512, 5.0, 201), (2048, 20.0, 1024), (64, 2.0, 32), (4096, 40.0, 2048), (4, 1.0, 4),
                                               (8, 2.0, 7), (128, 5.0, 128), (256, 5.0, 99), (8192, 40.0, 4001), (16384, 80.0, 8192),
                                               (32768, 100.0, 16001), (131072, 200.0, 65536),
                                               (262144, 400.0, 70000)])   # (beyond 16384: the spectra go through an HBM workspace; 262144 = BH_RF_MAX_NSAMP, the largest served)
@pytest.mark.parametrize("waveno", [0, 1])
def test_random_ragged_models_match_oracle(engine, oracle, nsamp, fsamp, nkeep, waveno):
    rs = np.random.RandomState(nsamp + waveno)
    nlay, h, vp, vs, rho = synth_models(rs, 70 if nsamp <= 16384 else (9 if nsamp < 262144 else 3), 21, lvz_frac=0.3, ragged=True)
    rf = engine.rf_batch(nlay, h, vp, vs, rho, 6.4, 2.5, nsamp, fsamp, 5.0, waveno, nkeep)
    orf = oracle.rf_batch(nlay, h.T, vp.T, vs.T, rho.T, 6.4, 2.5, nsamp, fsamp, 5.0, waveno, nkeep)
    peak = np.abs(orf).max(axis=1, keepdims=True)
    assert np.max(np.abs(rf - orf) / peak) <= TOL


@pytest.mark.parametrize("axis", ["n201", "n1024"])
def test_reference_golden_vectors(engine, axis):
    g = golden("rf_golden.npz")
    tx = g["x_" + axis]
    nsamp = 2 ** int(np.ceil(np.log2(tx.size * 2)))
    fsamp = 1.0 / float(np.round(tx[1] - tx[0], 4))
    for ic, (gauss, p) in enumerate(g["gauss_p"]):
        for iw in range(2):
            rf = engine.rf_batch(g["nlay"], g["h"], g["vp"], g["vs"], g["rho"], p, gauss, nsamp, fsamp, -tx[0], iw, tx.size,
                                 layout="model_major")
            ref = g["y_" + axis][:, ic, iw]
            assert np.max(np.abs(rf - ref)) <= 1e-4 * np.abs(ref).max()   # north_star bar
            assert np.max(np.abs(rf - ref)) <= TOL * np.abs(ref).max()     # what is achieved


@pytest.mark.parametrize("ref,waveno", [("prf", 0), ("srf", 1)])
def test_tutorial_files(engine, ref, waveno):
    x, y = st3(ref)
    h = np.array([[5., 23., 8., 0.]]).T; vs = np.array([[2.7, 3.6, 3.8, 4.4]]).T; vp = vs * 1.73
    rf = engine.rf_batch(np.array([4]), h, vp, vs, vp * 0.32 + 0.77, 6.4, 1.0, 512, 5.0, 5.0, waveno, 201)
    assert np.max(np.abs(rf[0] - y)) <= 1e-4


def test_quality_factors_and_nsv(engine, oracle):
    rs = np.random.RandomState(9)
    nlay, h, vp, vs, rho = synth_models(rs, 6, 8)
    qp = rs.uniform(100, 900, h.shape); qs = rs.uniform(50, 400, h.shape)
    rf = engine.rf_batch(nlay, h, vp, vs, rho, 5.0, 1.5, 512, 5.0, 5.0, 0, 300, qp=qp, qs=qs, nsv=3.1)
    for b in range(6):
        z = np.concatenate(([0], np.cumsum(h[:, b])[:-1]))
        k = vp[0, b] / vs[0, b]
        o = oracle.synrf(z, vp[:, b], vs[:, b], rho[:, b], qp[:, b], qs[:, b], 5.0, 1.5, 512, 5.0, 5.0, 3.1,
                         (2 - k ** 2) / (2 - 2 * k ** 2), "P")[2][:300]
        assert np.max(np.abs(rf[b] - o)) <= TOL * np.abs(o).max()


def test_spectral_cutoff_changes_nothing_and_nonfinite_models_stay_nonfinite(engine, oracle):
    """bins the Gauss low-pass puts below 1e-30 are not computed (rf_kernel.hip).  (1) With and without the
    cut-off (experiment switch rf_no_cut) the traces agree to 1e-13 of the peak on random models, for the c3 filter (a third of the bins
    cut) and the tutorial's (three quarters).  (2) The reference computes every bin, so a model whose coefficients are
    not finite gives an all-NaN trace there; the cut-off must not turn it into a finite one: same NaN rows as the oracle."""
    rs = np.random.RandomState(77)
    nlay, h, vp, vs, rho = synth_models(rs, 64, 12, lvz_frac=0.3, ragged=True)
    for gauss, nsamp, fsamp, nkeep in ((2.5, 2048, 20.0, 1024), (1.0, 512, 5.0, 201)):
        engine.set_tuning("rf_no_cut", 0)
        cut = engine.rf_batch(nlay, h, vp, vs, rho, 6.4, gauss, nsamp, fsamp, 5.0, 0, nkeep)
        engine.set_tuning("rf_no_cut", 1)       # (an experiment switch of the library, csrc/bh_tuning.h)
        try:
            full = engine.rf_batch(nlay, h, vp, vs, rho, 6.4, gauss, nsamp, fsamp, 5.0, 0, nkeep)
        finally:
            engine.set_tuning("rf_no_cut", 0)
        assert np.isfinite(full).all()
        assert np.max(np.abs(cut - full) / np.abs(full).max(axis=1, keepdims=True)) <= 1e-13
    # broken models: an infinite velocity, a NaN density, a zero S velocity in the crust
    nlay, h, vp, vs, rho = synth_models(rs, 8, 6)
    vp[2, 1] = np.inf
    rho[3, 3] = np.nan
    vs[1, 5] = 0.0
    rf = engine.rf_batch(nlay, h, vp, vs, rho, 6.4, 2.5, 2048, 20.0, 5.0, 0, 1024)
    orf = oracle.rf_batch(nlay, h.T, vp.T, vs.T, rho.T, 6.4, 2.5, 2048, 20.0, 5.0, 0, 1024)
    bad_o = ~np.isfinite(orf).all(axis=1)
    bad_e = ~np.isfinite(rf).all(axis=1)
    assert bad_o[[1, 3]].all() and np.array_equal(bad_e, bad_o), (bad_e, bad_o)
    ok = ~bad_o
    assert np.max(np.abs(rf[ok] - orf[ok]) / np.abs(orf[ok]).max(axis=1, keepdims=True)) <= TOL


def test_real_coefficient_recursion_equals_the_general_one(engine, oracle, monkeypatch):
    """Where no wave is post-critical at any interface the interface matrices are real and the synthesis kernel multiplies
    by real 2x2 matrices (rf_one_frequency<true>); BH_RF_NO_REALC forces the general form.  Same traces to 1e-13 of the
    peak; and a ray parameter beyond the critical one for the fast layers (complex matrices: the flag must say so and the
    general form run) still matches the oracle."""
    rs = np.random.RandomState(123)
    nlay, h, vp, vs, rho = synth_models(rs, 48, 12, lvz_frac=0.3, ragged=True)
    for p_ray, waveno in ((6.4, 0), (6.4, 1), (14.0, 0), (14.0, 1)):     # 14 s/deg = 0.126 s/km: evanescent P above 7.9 km/s
        monkeypatch.delenv("BH_RF_NO_REALC", raising=False)
        fast = engine.rf_batch(nlay, h, vp, vs, rho, p_ray, 2.0, 1024, 10.0, 5.0, waveno, 512)
        monkeypatch.setenv("BH_RF_NO_REALC", "1")
        gen = engine.rf_batch(nlay, h, vp, vs, rho, p_ray, 2.0, 1024, 10.0, 5.0, waveno, 512)
        monkeypatch.delenv("BH_RF_NO_REALC")
        orf = oracle.rf_batch(nlay, h.T, vp.T, vs.T, rho.T, p_ray, 2.0, 1024, 10.0, 5.0, waveno, 512)
        ok = np.isfinite(orf).all(axis=1)
        assert ok.sum() >= 20 and np.array_equal(np.isfinite(fast).all(axis=1), ok)
        peak = np.abs(orf[ok]).max(axis=1, keepdims=True)
        assert np.max(np.abs(fast[ok] - gen[ok]) / peak) <= 1e-13
        assert np.max(np.abs(fast[ok] - orf[ok]) / peak) <= TOL


def test_bad_arguments_fail_loudly(engine):
    from bayhunter_amd.engine import EngineError
    nlay, h, vp, vs, rho = synth_models(np.random.RandomState(1), 2, 3)
    with pytest.raises(EngineError):
        engine.rf_batch(nlay, h, vp, vs, rho, 6.4, 1.0, 500, 5.0, 5.0, 0, 100)  # nsamp not 2^k
    with pytest.raises(EngineError):   # 2^18 samples is the longest trace (rfmini has no cap)
        engine.rf_batch(nlay, h, vp, vs, rho, 6.4, 1.0, 1 << 19, 5.0, 5.0, 0, 100)
