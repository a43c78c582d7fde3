"""BASELINE.json's full sizes (batch = 4096, 10 layers, 30 periods, nsamp 2048): properties
that do not need the (slow) oracle -- determinism, permutation equivariance, duplicate
consistency, host-API == device-API -- plus an oracle spot check on a sample."""
import numpy as np
import pytest

from bayhunter_amd import engine as E
from bayhunter_amd.synth import synth_models, SWD_PERIODS, RF_TIME

pytestmark = pytest.mark.gpu
B = 4096


@pytest.fixture(scope="module")
def batch():
    rs = np.random.RandomState(20260927)
    return synth_models(rs, B, 10)


def test_swd_properties_full_batch(engine, oracle, batch):
    nlay, h, vp, vs, rho = batch
    perm = np.random.RandomState(1).permutation(B)
    for iwave, igr in ((2, 0), (1, 0), (2, 1), (1, 1)):
        v1, e1 = engine.swd_batch(nlay, h, vp, vs, rho, SWD_PERIODS, iwave, igr)
        v2, e2 = engine.swd_batch(nlay, h, vp, vs, rho, SWD_PERIODS, iwave, igr)
        assert np.array_equal(v1, v2) and np.array_equal(e1, e2)                 # deterministic
        v3, e3 = engine.swd_batch(nlay[perm], h[:, perm], vp[:, perm], vs[:, perm], rho[:, perm], SWD_PERIODS, iwave, igr)
        assert np.array_equal(v3, v1[perm]) and np.array_equal(e3, e1[perm])     # lane placement irrelevant
        ok = e1 == 0
        assert ok.mean() > 0.95
        assert np.all(np.isfinite(v1[ok])) and np.all(v1[ok] > 1.0) and np.all(v1[ok] < 5.0)
        # float32-rounded outputs, like `cg(k) = sngl(c(k))` (surfdisp96.f:298-303)
        assert np.array_equal(v1[ok], v1[ok].astype(np.float32).astype(np.float64))
        if igr == 0:  # phase velocity bounded by the extremal shear velocities
            assert np.all(v1[ok].max(axis=1) <= vs.max(axis=0)[ok] + 1e-6)
        idx = np.arange(0, B, 64)  # oracle spot check: one model of every wavefront
        ov, oe, _ = oracle.swd_batch(nlay[idx], h[:, idx].T, vp[:, idx].T, vs[:, idx].T, rho[:, idx].T, SWD_PERIODS, iwave, igr)
        assert np.array_equal(e1[idx], oe)
        assert np.max(np.abs(v1[idx][oe == 0] - ov[oe == 0]) / ov[oe == 0]) <= 1e-5


def test_rf_properties_full_batch(engine, oracle, batch):
    nlay, h, vp, vs, rho = batch
    args = (6.4, 2.5, 2048, 20.0, 5.0, 0, 1024)
    r1 = engine.rf_batch(nlay, h, vp, vs, rho, *args)
    r2 = engine.rf_batch(nlay, h, vp, vs, rho, *args)
    assert np.array_equal(r1, r2)
    perm = np.random.RandomState(2).permutation(B)
    r3 = engine.rf_batch(nlay[perm], h[:, perm], vp[:, perm], vs[:, perm], rho[:, perm], *args)
    assert np.array_equal(r3, r1[perm])
    assert np.all(np.isfinite(r1))
    # causality: before the direct arrival (t < -1.5 s; time origin at sample 100) only the
    # Gaussian filter's leakage remains, a few percent of the peak at most
    peak = np.abs(r1).max(axis=1)
    assert np.all(np.abs(r1[:, :70]).max(axis=1) <= 0.05 * peak) and np.all(peak > 0.01)
    idx = np.arange(0, B, 256)
    o = oracle.rf_batch(nlay[idx], h[:, idx].T, vp[:, idx].T, vs[:, idx].T, rho[:, idx].T, *args)
    assert np.max(np.abs(r1[idx] - o)) <= 1e-9 * np.abs(o).max()


def test_device_api_equals_host_api(engine, batch):
    import torch
    nlay, h, vp, vs, rho = batch
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_nlay, d_h, d_vp, d_vs, d_rho, d_per = t(nlay), t(h), t(vp), t(vs), t(rho), t(SWD_PERIODS)
    d_vel = torch.zeros((B, 30), dtype=torch.float64, device=dev)
    d_err = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    engine.swd_batch_dev(B, 10, d_nlay.data_ptr(), d_h.data_ptr(), d_vp.data_ptr(), d_vs.data_ptr(), d_rho.data_ptr(),
                         B, 1, 30, d_per.data_ptr(), 2, 0, d_vel.data_ptr(), d_err.data_ptr(), stream=st)
    torch.cuda.synchronize()
    v, e = engine.swd_batch(nlay, h, vp, vs, rho, SWD_PERIODS, 2, 0)
    assert np.array_equal(d_vel.cpu().numpy(), v) and np.array_equal(d_err.cpu().numpy(), e)


@pytest.mark.parametrize("hint", [0, 2, 4, 7, 12, 40])
def test_ragged_device_batch_any_depth_hint(engine, oracle, hint):
    """A transdimensional batch (2..21 layers, mostly shallow) through the device API with the caller's
    typical-depth hint right, wrong or absent: the hint only sizes lane groups and LDS rows (processing
    order by depth, two depth classes in one launch) -- velocities and flags are those of the oracle."""
    import torch
    rs = np.random.RandomState(17)
    Bn, L = 2600, 21   # (above 2048 models per call: below, every model has a wavefront of its own and the engine ignores the hint)
    nlay, h, vp, vs, rho = synth_models(rs, Bn, L, lvz_frac=0.2, ragged=True)
    shallow = rs.rand(Bn) < 0.8                           # 80 % of the models keep at most 7 layers
    for b in np.flatnonzero(shallow):
        n = min(int(nlay[b]), int(rs.randint(2, 8)))
        nlay[b] = n
        h[n - 1:, b] = 0.0
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    per = np.linspace(2, 50, 20)
    d = [t(a) for a in (nlay, h, vp, vs, rho, per)]
    d_vel = torch.zeros((Bn, per.size), dtype=torch.float64, device=dev)
    d_err = torch.zeros(Bn, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    try:
        engine.set_typical_layers(hint)
        for iwave, igr in ((2, 0), (1, 1)):
            engine.swd_batch_dev(Bn, L, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                 Bn, 1, per.size, d[5].data_ptr(), iwave, igr, d_vel.data_ptr(), d_err.data_ptr(), stream=st)
            torch.cuda.synchronize()
            ov, oe, _ = oracle.swd_batch(nlay, h.T, vp.T, vs.T, rho.T, per, iwave, igr)
            assert np.array_equal(d_err.cpu().numpy(), oe) and np.array_equal(d_vel.cpu().numpy(), ov), (hint, iwave, igr)
    finally:
        engine.set_typical_layers(0)


@pytest.mark.parametrize("search", ["fast", "reference"])
@pytest.mark.parametrize("workload", ["c2", "c3"])
def test_the_bench_path_against_the_oracle(workload, search):
    """The exact composition bench.py times (BASELINE configs[1] / [2]): `build_workload` -> `observed_data` ->
    `set_targets` -> `evaluate_batch_dev` on HBM-resident layer-major arrays, device pointers for every output,
    B = 4096, on a fresh engine -- 128 models spread over the batch against the oracle.  With the reference search: logL,
    misfits and failure flags against the oracle's dense restatement of Targets.py:314-347, at 1e-8 relative.  With the
    engine's DEFAULT search (the short refinement): failure flags equal, velocities within north_star's 1e-5 of the
    reference sequence's (1.2e-6 reached), the RF within 1e-4 of its peak, and logL = the oracle's dense likelihood of the
    device's own synthetics at 1e-8."""
    import torch
    import bench
    eng = E.Engine(0)
    assert eng.swd_search() == "fast"          # a fresh engine: the short refinement with its guard
    eng.set_swd_search(search)
    B, L = 4096, 10
    spec, batches, noise, truth, nrs = bench.build_workload(workload, B, L, seed=20260927)
    bench.observed_data(eng, spec, truth, nrs)
    eng.set_targets(spec)
    nt = len(spec)
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in batches[1]]
    d_noise = torch.from_numpy(noise).to(dev)
    d_logL = torch.zeros(B, dtype=torch.float64, device=dev)
    d_misf = torch.zeros((B, nt + 1), dtype=torch.float64, device=dev)
    d_err = torch.zeros(B, dtype=torch.int32, device=dev)
    eng.evaluate_batch_dev(B, L, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                           B, 1, d_noise.data_ptr(), d_logL.data_ptr(), d_misf.data_ptr(), d_err.data_ptr(),
                           stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    chk = bench.parity_check(spec, batches[1], noise, d_logL, d_misf, d_err, n=128)
    assert chk["n"] == 128 and chk["failure_flags_equal"]
    if search == "reference":
        assert chk["max_rel_logL"] <= 1e-8 and chk["max_rel_misfit"] <= 1e-8, chk
    d_ymod = torch.zeros((B, eng.ldy), dtype=torch.float64, device=dev)
    eng.evaluate_batch_dev(B, L, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                           B, 1, d_noise.data_ptr(), d_logL.data_ptr(), d_misf.data_ptr(), d_err.data_ptr(), ymod=d_ymod.data_ptr(),
                           stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    syn = bench.synthetics_check(spec, batches[1], noise, d_ymod, d_logL, d_err, n=128)
    assert syn["max_rel_velocity"] <= (0.0 if search == "reference" else 2e-6), syn
    assert syn["max_rel_logL_of_the_device_synthetics"] <= 1e-8, syn
    assert syn.get("max_rf_over_peak", 0.0) <= 1e-8, syn
    assert int((d_err != 0).sum().item()) < B // 20 and bool(torch.isfinite(d_logL).all().item())
    eng.close()


def test_large_call_of_the_lane_kernel_two_wavefronts_per_workgroup(engine, oracle):
    """More than 2048 wavefronts in one call: the lane-per-evaluation kernel runs with two wavefronts per workgroup
    (shared libm tables) and time-sliced priorities, the Neville orders from five on in its global work array.
    40 000 models x 4 trial lanes: identical to the planner's own choice for the batch, bit-identical to the oracle on
    a sample (incl. the models with the longest runs of interpolation steps: low-velocity zones)."""
    rs = np.random.RandomState(424242)
    Bb = 40000
    nlay, h, vp, vs, rho = synth_models(rs, Bb, 10, lvz_frac=0.3, ragged=True)
    try:
        for iwave, igr in ((2, 0), (1, 1)):
            engine.set_swd_group(0); engine.set_swd_lookahead(0)
            v0, e0 = engine.swd_batch(nlay, h, vp, vs, rho, SWD_PERIODS, iwave, igr)
            engine.set_swd_group(1); engine.set_swd_lookahead(4)          # 2500 wavefronts
            v1, e1 = engine.swd_batch(nlay, h, vp, vs, rho, SWD_PERIODS, iwave, igr)
            assert np.array_equal(v0, v1) and np.array_equal(e0, e1)
            idx = np.arange(0, Bb, 97)
            ov, oe, _ = oracle.swd_batch(nlay[idx], h[:, idx].T, vp[:, idx].T, vs[:, idx].T, rho[:, idx].T, SWD_PERIODS, iwave, igr)
            assert np.array_equal(e1[idx], oe) and np.array_equal(v1[idx], ov)
    finally:
        engine.set_swd_group(0); engine.set_swd_lookahead(0)
