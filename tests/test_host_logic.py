"""Host-side logic of the plugin mirror (no GPU): model parametrisation, sampling parameters,
covariance-law selection, dense-vs-closed-form likelihood identities."""
import numpy as np
import pytest

from conftest import golden
import bayhunter_amd as bh
from bayhunter_amd import Targets


def test_get_vp_vs_h_matches_reference():
    g = golden("model_golden.npz")
    L = g["nuclei"].shape[1] // 2
    for i in range(g["n"].size):
        n = g["n"][i]
        model = np.concatenate((g["nuclei"][i, :n], g["nuclei"][i, L:L + n]))
        mantle = list(g["mantle"]) if g["use_mantle"][i] else None
        vp, vs, h = bh.Model.get_vp_vs_h(model, g["vpvs"][i], mantle)
        assert np.array_equal(vp, g["vp"][i, :n]) and np.array_equal(vs, g["vs"][i, :n])
        assert np.array_equal(h, g["h"][i, :n])
        # NaN-padded storage form (SingleChain.py:207-241) splits the same way
        assert bh.Model.split_modelparams(g["nuclei"][i])[0] == n


def test_pack_batch_layout():
    g = golden("model_golden.npz")
    L = g["nuclei"].shape[1] // 2
    models = [np.concatenate((g["nuclei"][i, :g["n"][i]], g["nuclei"][i, L:L + g["n"][i]])) for i in range(6)]
    nlay, h, vp, vs = bh.Model.pack_batch(models, g["vpvs"][:6])
    assert h.shape == (nlay.max(), 6)
    for b in range(6):
        assert np.array_equal(vp[:nlay[b], b], g["vs"][b, :nlay[b]] * g["vpvs"][b])
        assert np.all(h[nlay[b]:, b] == 0)


def test_rf_sampling_parameters():
    p = bh.RFminiModRF(np.linspace(-5, 35, 201), "prf")
    assert (p.fsamp, p.tshft, p.nsamp) == (5.0, 5.0, 512)
    p = bh.RFminiModRF(-5 + 0.05 * np.arange(1024), "srf")
    assert (round(p.fsamp, 9), p.tshft, p.nsamp) == (20.0, 5.0, 2048)
    assert p.modelparams["wtype"] == "SV"
    with pytest.raises(ValueError):
        bh.RFminiModRF(np.array([0., 1., 2.5]), "prf")
    # the engine's cap on the transform length (the reference has none, rfmini_modrf.py:62): named before any native call
    assert bh.RFminiModRF(-5 + 0.01 * np.arange(8193), "prf").nsamp == 32768        # (beyond 16384: the kernel's HBM workspace path)
    assert bh.RFminiModRF(-5 + 0.01 * np.arange(131072), "prf").nsamp == 262144 == bh.RFminiModRF.MAX_NSAMP
    with pytest.raises(ValueError, match="MAX_NSAMP = 262144"):
        bh.RFminiModRF(-5 + 0.01 * np.arange(131073), "prf")


def test_surfdisp_tags_and_resampling_grid():
    assert bh.SurfDisp(np.arange(1, 5.), "rdispgr").get_surftags("ldispph") == (1, 0)
    with pytest.raises(ReferenceError):
        bh.SurfDisp(np.arange(1, 5.), "xdisp")
    s = bh.SurfDisp(np.linspace(1.5, 70, 80), "rdispph")
    assert s.obsx_int.size == 60 and s.obsx_int[0] == 1.5 and s.obsx_int[-1] == 70


def test_noise_law_selection_rules():
    x = np.linspace(1, 30, 10); y = np.ones(10)
    tx = np.linspace(-5, 35, 201)
    t_sw = bh.RayleighDispersionPhase(x, y)
    t_sw_err = bh.RayleighDispersionPhase(x, y, yerr=np.full(10, 0.1))
    t_rf = bh.PReceiverFunction(tx, np.zeros(201))
    t_rf2 = bh.PReceiverFunction(tx, np.zeros(201))
    t_sw2 = bh.LoveDispersionPhase(x, y)
    bh.select_noise_laws([t_sw, t_sw_err, t_rf, t_rf2, t_sw2], corrfix=[True, True, True, False, True],
                         noise_corr=[0, 0, 0.9, (0.2, 0.9), 0.5], rcond=1e-6)
    assert [t.noise_law for t in (t_sw, t_sw_err, t_rf, t_rf2, t_sw2)] == \
        ["nocorr", "nocorr_scalederr", "gauss", "exp", "exp"]
    assert t_rf.valuation.corr_inv.shape == (201, 201)


def test_speculation_depth_stays_in_the_latency_regime():
    """device_chains.auto_spec_depth: chains x (2^d - 1) evaluations per launch within the budget, 1 <= d <= 7."""
    from bayhunter_amd.device_chains import auto_spec_depth
    assert [auto_spec_depth(c, 1024) for c in (1, 8, 9, 64, 68, 146, 341, 342, 4096)] == [7, 7, 6, 4, 4, 3, 2, 1, 1]
    assert auto_spec_depth(8, 512) == 6 and auto_spec_depth(8, 0) == 1 and auto_spec_depth(64, 2048) == 5
    for c in (1, 3, 8, 64, 500):
        d = auto_spec_depth(c, 1024)
        assert d == 7 or c * ((1 << (d + 1)) - 1) > 1024
        assert d == 1 or c * ((1 << d) - 1) <= 1024


def test_law_is_read_back_from_the_installed_accessor():
    """SingleChain.py:159-205 assigns target.get_covariance; SingleTarget.law() identifies it (host logic only)."""
    import copy
    import pickle
    x = np.linspace(1, 30, 10); y = np.ones(10)
    t = bh.RayleighDispersionPhase(x, y)
    with pytest.raises(RuntimeError):
        t.law()
    v = t.valuation
    for acc, law in ((v.get_covariance_nocorr, "nocorr"), (v.get_covariance_nocorr_scalederr, "nocorr_scalederr"),
                     (v.get_covariance_exp, "exp")):
        t.get_covariance = acc
        assert t.law() == law and t.noise_law == law
    t.get_covariance = v.get_covariance_gauss
    with pytest.raises(RuntimeError):           # Gauss accessor without init_covariance_gauss
        t.law()
    v.init_covariance_gauss(0.9, 10, rcond=1e-6)
    assert t.law() == "gauss" and t.engine_desc()["rinv"].shape == (10, 10)
    v.init_covariance_gauss(0.9, 7)
    with pytest.raises(ValueError):             # R^-1 of another size
        t.law()
    v.init_covariance_gauss(0.9, 10)
    # the reset before pickling keeps the law; a deep copy / pickle round trip rebinds the accessor to the copy
    t2 = pickle.loads(pickle.dumps(t))
    assert t2.law() == "gauss" and t2.get_covariance.__self__ is t2.valuation
    t.get_covariance = None
    assert t.law() == "gauss"
    t3 = copy.deepcopy(t)
    assert t3.get_covariance is None and t3.law() == "gauss"
    # foreign callables and subclass overrides are errors, not nocorr
    t.get_covariance = lambda sigma, size, yerr=None, corr=0: (np.eye(size), 0.0)
    with pytest.raises(TypeError):
        t.law()

    class MyValuation(bh.Valuation):
        def get_covariance_exp(self, corr, sigma, size, yerr=None):
            return np.eye(size), 0.0
    t.valuation = MyValuation()
    t.get_covariance = t.valuation.get_covariance_exp
    with pytest.raises(TypeError):
        t.law()
    t.get_covariance = t.valuation.get_covariance_nocorr     # inherited, not overridden
    assert t.law() == "nocorr"


@pytest.mark.parametrize("n", [1, 2, 7, 64])
def test_closed_forms_equal_dense_forms(oracle, n):
    """App. C of SURVEY.md: the O(n) expressions the engine uses == the reference's dense ones."""
    rs = np.random.RandomState(n)
    d = rs.normal(0, 0.1, n); yobs = rs.normal(0, 1, n); ymod = yobs + d
    r, s = 0.63, 0.07
    v = Targets.Valuation()
    if n > 1:
        c_inv, ld = v.get_covariance_exp(r, s, n)
        dense = v.get_likelihood(yobs, ymod, c_inv, ld)
        assert abs(dense - oracle.loglike_dense(2, ymod, yobs, r, s)) <= 1e-10 * abs(dense)
    edge = d[0] ** 2 + d[-1] ** 2 if n > 1 else d[0] ** 2
    phi = ((1 + r * r) * np.sum(d * d) - r * r * edge - 2 * r * np.sum(d[:-1] * d[1:])) / (s * s * (1 - r * r))
    closed = -0.5 * (n * np.log(2 * np.pi) + 2 * n * np.log(s) + (n - 1) * np.log(1 - r * r)) - 0.5 * phi
    assert abs(closed - oracle.loglike_dense(2, ymod, yobs, r, s)) <= 1e-9 * abs(closed)
    yerr = rs.uniform(0.01, 0.05, n)
    c_inv, ld = v.get_covariance_nocorr_scalederr(s, n, yerr)
    assert abs(v.get_likelihood(yobs, ymod, c_inv, ld) - oracle.loglike_dense(1, ymod, yobs, 0, s, yerr=yerr)) <= 1e-9 * abs(ld)


def test_philox_known_answers():
    """The numpy Philox4x32-10 used to check the device chain step reproduces the Random123
    known-answer vectors (kat_vectors: philox4x32 10)."""
    from philox_ref import philox4x32_10
    z = philox4x32_10(np.zeros((1, 4), dtype=np.uint32), (0, 0))[0]
    assert [hex(int(v)) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = philox4x32_10(np.full((1, 4), 0xFFFFFFFF, dtype=np.uint32), (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [hex(int(v)) for v in f] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    p = philox4x32_10(np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], dtype=np.uint32),
                      (0xa4093822, 0x299f31d0))[0]
    assert [hex(int(v)) for v in p] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_weighted_values_repeat_rows_like_the_reference():
    """ModelMatrix.get_weightedvalues (src/Models.py:227-274): rows repeated by their weights; expected values
    written out by hand (the build container also compared with the reference's own loops)."""
    from bayhunter_amd import ModelMatrix
    w = [2, 0, 3]
    models = np.arange(12.).reshape(3, 4)
    likes = np.array([1., 2., 3.])
    misfits = np.arange(6.).reshape(3, 2)
    noise = misfits + 10
    vpvs = np.array([1.7, 1.8, 1.9])
    wm, wl, wmis, wn, wv = ModelMatrix.get_weightedvalues(w, models, likes, misfits, noise, vpvs)
    rows = [0, 0, 2, 2, 2]
    assert np.array_equal(wm, models[rows]) and np.array_equal(wl, likes[rows]) and np.array_equal(wmis, misfits[rows])
    assert np.array_equal(wn, noise[rows]) and np.array_equal(wv, vpvs[rows])
    assert ModelMatrix.get_weightedvalues(w, likes=likes)[0] is None
    assert np.array_equal(ModelMatrix.get_weightedvalues(w, misfits=[1., 2., 3.])[2], np.array([1., 1., 3., 3., 3.]))


def test_bench_line_stays_compact_and_parseable():
    """bench.py prints ONE line the driver parses out of an 8 KB tail of stdout (BENCH_r04: a 24.6 KB line was not parsed):
    the line builder on a canned full record (round 4's 24.6 KB record, blown up further) stays under 8000 bytes and keeps
    the contract's keys, `roofline` and `cpu_baseline`."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    full = json.load(open(os.path.join(root, "profiles", "r04_bench_default_final.json")))
    full["some_future_block"] = {"k%d" % i: "x" * 100 for i in range(500)}       # whatever the blocks grow by stays out
    full["cpu_baseline"]["sample"] = "y" * 5000
    text = bench.compact_line(full, "bench_full.json")
    assert "\n" not in text and len(text) < bench.LINE_LIMIT and len(text) < 4096
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_check", "summary", "full_record"):
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-5) and line["config"]["workload"].startswith("joint Rayleigh+Love")
    assert line["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["binding"]["frac"] > 0
    assert line["cpu_baseline"]["cores"] == 16 and line["cpu_baseline"]["kind"] == "reference"
    assert len(line["parity_check"]) <= 6 and line["parity_check"]["failure_flags_equal"] is True
    # an N > 1 line carries the collective check and the per-rank spread and obeys the same limit
    full["collective_check"] = {"backend": "nccl", "world_size": 8, "ranks_seen_by_all_reduce": 8, "rccl_ranks": 8, "devices": 8}
    full["rank_ms_per_step"] = {"min": 2.2, "max": 2.4, "all": [2.3] * 8}
    line8 = json.loads(bench.compact_line(full, None))
    assert line8["collective_check"]["rccl_ranks"] == 8 and "all" not in line8["rank_ms_per_step"]
    assert len(bench.compact_line(full, None)) < 4096
