"""Device-resident chain step (bayhunter_amd/device_chains.py, csrc/chain_kernel.hip) against the
host chain driver that replays the reference draw for draw (bayhunter_amd/chains.py, itself pinned to
recorded reference runs in test_gpu_chains.py).

The host driver's RandomState is replaced by an object handing out the SAME six draws per iteration
the device kernel uses (tests/philox_ref.py), so both must walk the same trajectory:
  * injected draws  -> the device reads them from a buffer: states must be identical;
  * device Philox   -> the host gets the numpy Philox restatement of the same draws (the Box-Muller
                       normal can differ in the last bit between numpy and the device): same decisions,
                       states equal to 1e-9.
"""
import os

import numpy as np
import pytest

from conftest import golden
import bayhunter_amd as bh
from bayhunter_amd.chains import ChainBatch
from bayhunter_amd.device_chains import DeviceChains
from philox_ref import InjectedRandomState, draws
from test_gpu_chains import SETUPS, make_targets

pytestmark = pytest.mark.gpu


def host_twin(dc, targets, init, priors):
    """A ChainBatch holding the device chains' current state, with injectable draws."""
    hb = ChainBatch(targets, list(range(dc.C)), init, priors)
    st = dc.state_host()
    for c, ch in enumerate(hb.chains):
        n = int(st["n"][c])
        ch.currentmodel = np.concatenate((st["vs"][:n, c], st["z"][:n, c]))
        ch.currentnoise = st["noise"][:, c].copy()
        ch.currentvpvs = float(st["vpvs"][c])
        ch.currentlikelihood = float(st["like"][c])
        ch.currentmisfits = st["misfits"][:, c].copy()
        ch.propdist = st["propdist"][:, c].copy()
        ch.rstate = InjectedRandomState()
    hb.iiter = dc.iiter
    return hb


def compare(dc, hb, exact):
    st = dc.state_host()
    eq = (lambda a, b: np.array_equal(a, b)) if exact else (lambda a, b: np.allclose(a, b, rtol=1e-9, atol=1e-12))
    for c, ch in enumerate(hb.chains):
        n = ch.currentmodel.size // 2
        assert st["n"][c] == n, (dc.iiter, c)
        assert eq(st["vs"][:n, c], ch.currentmodel[:n]) and eq(st["z"][:n, c], ch.currentmodel[n:]), (dc.iiter, c)
        assert eq(st["noise"][:, c], ch.currentnoise) and eq(st["vpvs"][c], ch.currentvpvs), (dc.iiter, c)
        assert eq(st["like"][c], ch.currentlikelihood), (dc.iiter, c, st["like"][c], ch.currentlikelihood)
        assert eq(st["misfits"][:, c], ch.currentmisfits)
        assert eq(st["propdist"][:, c], ch.propdist), (dc.iiter, c, st["propdist"][:, c], ch.propdist)
        assert np.array_equal(st["proposed"][:, c], ch.proposed) and np.array_equal(st["accepted"][:, c], ch.accepted)


@pytest.mark.parametrize("name,depth", [("exp", 1), ("gauss", 1), ("exp", 3), ("gauss", 4)])
def test_injected_draws_same_trajectory(name, depth):
    """depth > 1: speculative windows (the proposals of both outcomes of the next `depth` decisions, one evaluation
    launch, the realised path replayed) against the host twin iterating one step at a time."""
    g = golden("chain_golden.npz")
    su = SETUPS[name]
    init = dict(su["init"], iter_burnin=1100, iter_main=150, lvz=0.1, hvz=0.4)
    priors = dict(su["priors"], mantle=(4.3, 1.8))
    C = 48
    targets = make_targets(g)
    dc = DeviceChains(targets, C, init, priors, seed=7, inject=True, spec_depth=depth, search="reference")   # (the host twin's bits)
    hb = host_twin(dc, targets, init, priors)
    rs = np.random.RandomState(99)
    it = 0
    while dc.iiter < dc.iter_phase2:
        w = dc.window()
        d = np.zeros((depth, 6, C))
        for k in range(w):
            d[k] = np.vstack((rs.uniform(size=(5, C)), rs.normal(size=(1, C))))
        dc.t["inject"].copy_(dc.torch.from_numpy(d))
        dc.torch.cuda.synchronize()
        assert dc.iterate() == w
        for k in range(w):
            for c, ch in enumerate(hb.chains):
                ch.rstate.set(d[k][:, c])
            hb.iterate()
        it += 1
        if it % 50 == 0 or dc.iiter in (-999, 1):      # incl. right after the width adaptations
            compare(dc, hb, exact=True)
    assert dc.launches == it and (depth == 1 or it < 1250 / depth + 8)
    compare(dc, hb, exact=True)
    st = dc.state_host()
    assert (st["accepted"].sum(axis=0) > 50).all() and (st["n"] != st["n"][0]).any()   # chains really moved / differ
    if name == "exp":   # (with a fixed vp/vs that family is never proposed and the reference never adapts, :584)
        assert (st["propdist"] != np.asarray(hb.initparams["propdist"])[:, None]).any()    # widths were adapted


def test_device_philox_same_trajectory_and_reproducible():
    g = golden("chain_golden.npz")
    su = SETUPS["exp"]
    init = dict(su["init"], iter_burnin=300, iter_main=120)
    C, seed = 40, (0x1234 << 32) | 0x9abcdef1
    targets = make_targets(g)
    dc = DeviceChains(targets, C, init, su["priors"], seed=seed, search="reference")   # (the host twin evaluates with the reference's bits)
    hb = host_twin(dc, targets, init, su["priors"])
    while dc.iiter < dc.iter_phase2:
        before = dc.iiter
        for k in range(dc.window()):                    # (the default depth for 40 chains is 4 iterations per launch)
            d = draws(seed, C, dc.iiter + k)
            for c, ch in enumerate(hb.chains):
                ch.rstate.set(d[:, c])
            hb.iterate()
        dc.iterate()
        if dc.iiter // 60 != before // 60:
            compare(dc, hb, exact=False)
    compare(dc, hb, exact=False)
    a = dc.state_host()
    # same seed -> identical run; another seed -> another trajectory
    dc2 = DeviceChains(make_targets(g), C, init, su["priors"], seed=seed, search="reference").run()
    b = dc2.state_host()
    for k in ("n", "vs", "z", "like", "noise", "vpvs", "propdist", "accepted"):
        assert np.array_equal(a[k], b[k]), k
    dc3 = DeviceChains(make_targets(g), C, init, su["priors"], seed=seed + 1, search="reference").run()
    assert not np.array_equal(a["like"], dc3.state_host()["like"])


@pytest.mark.parametrize("name", ["exp", "gauss"])
def test_speculative_windows_walk_the_sequential_trajectory(name):
    """d iterations per evaluation launch must be the d = 1 run with the same seed, bit for bit --
    states, counters, adapted widths, snapshots -- for the exp and the Gauss set-up, across the adaptation
    iterations (-1000, 0), with 8 chains (BASELINE configs[3] per-GPU share) at several depths incl. the default."""
    g = golden("chain_golden.npz")
    su = SETUPS[name]
    init = dict(su["init"], iter_burnin=1150, iter_main=250, maxmodels=25)
    C = 8
    keys = ("n", "vs", "z", "like", "noise", "vpvs", "misfits", "propdist", "proposed", "accepted", "naccepted")

    def run(depth):
        dc = DeviceChains(make_targets(g), C, init, su["priors"], seed=20260928, spec_depth=depth).run()
        return dc, dc.state_host(), dc.samples("p1"), dc.samples("p2")

    d1, s1, p1a, p2a = run(1)
    assert d1.launches == 1400
    for depth in (2, 5, None):
        dk, sk, p1b, p2b = run(depth)
        assert dk.iiter == d1.iiter and dk.launches < 1400 / min(dk.depth, 4) * 1.05 + 10   # (+ windows cut at snapshots / adaptations)
        for k in keys:
            assert np.array_equal(s1[k], sk[k]), (depth, k)
        for a, b in ((p1a, p1b), (p2a, p2b)):
            for k in a:
                assert np.array_equal(a[k], b[k], equal_nan=True), (depth, k)
    assert dk.depth == 7 and (s1["accepted"].sum(axis=0) > 100).all()


def test_speculative_windows_with_tempering_stop_at_the_exchanges():
    """Windows end at temperature exchanges: a tempered run with 5 iterations per launch equals the
    one-iteration-per-launch run (betas, states, accepted swaps)."""
    g = golden("chain_golden.npz")
    su = SETUPS["exp"]
    init = dict(su["init"], iter_burnin=280, iter_main=60, maxmodels=10)
    nl, nr = 4, 4
    C = nl * nr
    ladder = np.repeat(np.arange(nl), nr)
    betas = np.tile(1.0 / np.geomspace(1.0, 20.0, nr), nl)
    out = []
    for depth in (1, 5):
        dc = DeviceChains(make_targets(g), C, init, su["priors"], seed=5, betas=betas, ladder=ladder, swap_every=7,
                          spec_depth=depth).run()
        out.append((dc.state_host(), dc.nswaps, dc.sweep, dc.launches))
    (a, na, sa, la), (b, nb, sb, lb) = out
    assert na == nb and sa == sb == 340 // 7 and na > 3 and lb < la / 2
    for k in ("n", "vs", "z", "like", "noise", "vpvs", "beta", "propdist", "accepted"):
        assert np.array_equal(a[k], b[k]), k


def test_many_chains_converge_and_save(tmp_path):
    """512 chains from random starts: all run the whole schedule on the device, the likelihood of
    (nearly) every chain climbs to the level the recorded reference chains reach, and the result
    files have the reference's names, dtypes and row layout."""
    g = golden("chain_golden.npz")
    su = SETUPS["exp"]
    init = dict(su["init"], iter_burnin=1500, iter_main=700, maxmodels=100, savepath=str(tmp_path))
    C = 512
    dc = DeviceChains(make_targets(g), C, init, su["priors"], seed=3).run()
    st = dc.state_host()
    ref_final = min(float(g["exp_s%d_likes" % s][-1]) for s in su["seeds"])
    # (the chains start at logL of order -1e4 .. -1e5; the two recorded reference chains end at 610 and 760)
    assert np.median(st["like"]) > ref_final - 150.0, (np.median(st["like"]), ref_final)
    assert np.mean(st["like"] > 0.0) > 0.9
    prop = st["proposed"].sum(axis=1) / st["proposed"].sum()
    assert (prop > 0.1).all()                                   # every family of moves is being proposed
    rate = st["accepted"].sum() / st["proposed"].sum()
    assert 0.15 < rate < 0.9, rate
    assert dc.thinning == 7 and len(dc.snap["p2"]) == 100
    s = dc.samples("p2")
    assert s["models"].shape == (100, C, 2 * dc.ML) and s["models"].dtype == np.float32
    path = dc.save()
    m = np.load(os.path.join(path, "c%.3d_p2models.npy" % (C - 1)))
    lk = np.load(os.path.join(path, "c%.3d_p2likes.npy" % (C - 1)))
    assert m.shape == (100, 2 * dc.ML) and lk.shape == (100,)
    n, vs, z = bh.Model.split_modelparams(m[-1])
    last = dc.snap["p2"][-1]                                     # the state the last file row was taken from
    assert n == last["n"][C - 1] and np.all(np.diff(z) > 0)
    assert np.array_equal(vs, last["vs"][:n, C - 1]) and np.array_equal(z, last["z"][:n, C - 1])


def test_parallel_tempering_on_device_chains():
    """Parallel tempering (no reference counterpart: invariants only).  32 ladders x 6 temperatures:
    beta = 1 everywhere and no swaps reproduces the untempered run bit for bit; with a ladder every ladder
    keeps its set of temperatures, swaps happen, hot chains accept more, and the chains that hold beta = 1
    sit at a higher likelihood than the hottest ones."""
    g = golden("chain_golden.npz")
    su = SETUPS["exp"]
    init = dict(su["init"], iter_burnin=900, iter_main=300, maxmodels=30)
    nl, nr = 32, 6
    C = nl * nr
    plain = DeviceChains(make_targets(g), C, init, su["priors"], seed=11).run().state_host()
    ones = DeviceChains(make_targets(g), C, init, su["priors"], seed=11, betas=np.ones(C), ladder=np.repeat(np.arange(nl), nr),
                        swap_every=0).run().state_host()
    for k in ("n", "vs", "z", "like", "noise", "vpvs", "propdist", "accepted"):
        assert np.array_equal(plain[k], ones[k]), k
    ladder = np.repeat(np.arange(nl), nr)
    betas = np.tile(1.0 / np.geomspace(1.0, 30.0, nr), nl)
    dc = DeviceChains(make_targets(g), C, init, su["priors"], seed=11, betas=betas, ladder=ladder, swap_every=20).run()
    st = dc.state_host()
    assert dc.sweep == 1200 // 20 and dc.nswaps > 50
    for lid in range(nl):
        assert np.allclose(np.sort(st["beta"][ladder == lid]), np.sort(betas[ladder == lid]), rtol=0, atol=0)
    cold, hot = st["beta"] == 1.0, st["beta"] == betas.min()
    assert cold.sum() == nl and hot.sum() == nl
    assert np.median(st["like"][cold]) > np.median(st["like"][hot]) + 20.0
    s = dc.samples("p2")
    assert s["beta"].shape == (len(dc.snap["p2"]), C) and (np.sum(s["beta"] == 1.0, axis=1) == nl).all()
    # acceptance is easier at high temperature: compare chains that spent the run hot vs cold on average
    mean_beta = np.mean([r["beta"] for r in dc.snap["p1"] + dc.snap["p2"]], axis=0)
    rate = st["accepted"].sum(axis=0) / np.maximum(st["proposed"].sum(axis=0), 1)
    assert np.mean(rate[mean_beta < 0.2]) > np.mean(rate[mean_beta > 0.6])


def test_exchange_on_the_device_equals_the_host_exchange(monkeypatch):
    """The swap sweep as torch operations on the engine's stream (parallel.DeviceExchange: no host synchronisation)
    takes the decisions of the NumPy form (parallel.ladder_swap_betas) bit for bit: same betas after every sweep,
    same chain states at the end, same number of accepted swaps -- with unequal ladders and tied temperatures."""
    g = golden("chain_golden.npz")
    su = SETUPS["exp"]
    init = dict(su["init"], iter_burnin=300, iter_main=100, maxmodels=20)
    rs = np.random.RandomState(5)
    sizes = [1, 2, 3, 5, 8, 4, 7, 6, 8, 8]
    ladder = rs.permutation(np.repeat(np.arange(len(sizes)), sizes))           # ladders scattered over the chains
    C = ladder.size
    betas = np.ones(C)
    for lid, n in enumerate(sizes):
        b = 1.0 / np.geomspace(1.0, 20.0, n)
        if n >= 5:
            b[2] = b[1]                                                           # a tie
        betas[ladder == lid] = b

    def run(host):
        monkeypatch.setenv("BH_PT_HOST_EXCHANGE", "1" if host else "0")
        dc = DeviceChains(make_targets(g), C, init, su["priors"], seed=77, betas=betas, ladder=ladder, swap_every=5)
        assert (dc._dev_exchange is None) == host
        trace = []
        while dc.iiter < dc.iter_phase2:
            dc.iterate()
            if dc.iiter % 5 == 0 and dc.iiter < 0:    # (burn-in: look after every sweep; main phase: the host runs ahead)
                trace.append(dc.t["beta"].cpu().numpy().copy())
        return dc, np.array(trace), dc.state_host()

    d1, t1, s1 = run(True)
    d2, t2, s2 = run(False)
    assert t1.shape == t2.shape and np.array_equal(t1, t2)
    assert d1.sweep == d2.sweep == 80 and d1.nswaps == d2.nswaps and d1.nswaps > 20
    for k in ("n", "vs", "z", "like", "noise", "vpvs", "beta"):
        assert np.array_equal(s1[k], s2[k]), k


def test_posterior_statistics_match_reference_order_chains():
    """Statistical parity (the accept decisions are chaotic, parity of the sampler with its own
    random stream can only be statistical): the same problem sampled by 64 reference-order chains
    (ChainBatch, numpy Mersenne Twister, pinned to the reference draw for draw) and by 64 device chains
    (Philox); posterior summaries must agree within Monte-Carlo error.  This test caught a normal deviate
    that shared Philox bits with the move choice -- the walk drifted to the prior bounds (20 sigma)."""
    g = golden("chain_golden.npz")
    su = SETUPS["exp"]
    N, burn, main = 64, 3000, 2000
    init = dict(su["init"], iter_burnin=burn, iter_main=main, acceptance=(40, 45), maxmodels=100)

    def summaries(models, likes, noise, vpvs):
        n = np.array([bh.Model.split_modelparams(m)[0] for m in models])
        depths = np.array([2.0, 10.0, 25.0, 40.0, 55.0])
        v = np.zeros((models.shape[0], depths.size))
        for i, m in enumerate(models):
            _, vs, z = bh.Model.split_modelparams(m)
            v[i] = vs[np.argmin(np.abs(z[:, None] - depths[None, :]), axis=0)]
        return np.concatenate(([likes.mean(), n.mean(), vpvs.mean(), noise[:, 1].mean(), noise[:, 2].mean(), noise[:, 3].mean()],
                               v.mean(axis=0)))

    hb = ChainBatch(make_targets(g), list(range(500, 500 + N)), init, su["priors"]).run()
    H = []
    for c in range(N):
        a = hb.chain_arrays(c)
        p2 = a["iters"] >= 0
        w = np.diff(np.concatenate((a["iters"][p2], [hb.iiter]))).astype(int)       # dwell times = weights
        rep = lambda k: np.repeat(a[k][p2], w, axis=0)[::20]
        H.append(summaries(rep("models"), rep("likes"), rep("noise"), rep("vpvs")))
    H = np.array(H)
    dc = DeviceChains(make_targets(g), N, init, su["priors"], seed=4242).run()
    s = dc.samples("p2")
    D = np.array([summaries(s["models"][:, c], s["likes"][:, c], s["noise"][:, c], s["vpvs"][:, c]) for c in range(N)])
    sem = np.sqrt(H.var(axis=0, ddof=1) / N + D.var(axis=0, ddof=1) / N)
    z = (D.mean(axis=0) - H.mean(axis=0)) / sem
    assert np.all(np.abs(z) < 4.5), z


def test_search_modes_sample_the_same_posterior():
    """The chains' default search ("fast": roots within 1.2e-6 of the reference's instead of its bits) does
    not change what they sample: the same problem sampled by 768 device chains with the reference's sequence and by 768
    others (other seeds) with the default; posterior summaries -- logL, number of nuclei, vp/vs, the three free noise
    parameters, vs at five depths -- agree within Monte-Carlo error (pairs of runs of ONE mode scatter up to 2.6 standard
    errors, profiles/r03_chains_stat_search.txt; the pooled difference there was 1.7)."""
    g = golden("chain_golden.npz")
    N, burn, main = 256, 4000, 2000
    priors = dict(vpvs=(1.4, 2.1), layers=(1, 10), vs=(2, 5), z=(0, 60), rfnoise_corr=(0.35, 0.75), rfnoise_sigma=(1e-5, 0.05),
                  swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.1))
    init = dict(nchains=1, iter_burnin=burn, iter_main=main, acceptance=(40, 45), thickmin=0.1, lvz=0.1, hvz=None, rcond=None,
                maxmodels=main // 20)

    def targets():
        t1 = bh.RayleighDispersionPhase(g["xsw"], g["ysw"])
        t2 = bh.PReceiverFunction(g["xrf"], g["yrf"])
        t2.moddata.plugin.set_modelparams(gauss=1.0, p=6.4)
        return bh.JointTarget([t1, t2])

    def summaries(models, likes, noise, vpvs):
        n = np.array([bh.Model.split_modelparams(m)[0] for m in models])
        depths = np.array([2.0, 10.0, 25.0, 40.0, 55.0])
        v = np.zeros((models.shape[0], depths.size))
        for i, m in enumerate(models):
            _, vs, z = bh.Model.split_modelparams(m)
            v[i] = vs[np.argmin(np.abs(z[:, None] - depths[None, :]), axis=0)]
        return np.concatenate(([likes.mean(), n.mean(), vpvs.mean(), noise[:, 1].mean(), noise[:, 2].mean(), noise[:, 3].mean()],
                               v.mean(axis=0)))

    out = {"reference": [], "fast": []}
    for search, seeds in (("reference", (77, 79, 81)), ("fast", (83, 85, 87))):
        for seed in seeds:
            dc = DeviceChains(targets(), N, init, priors, seed=seed, search=search).run()
            s = dc.samples("p2")
            out[search] += [summaries(s["models"][:, c], s["likes"][:, c], s["noise"][:, c], s["vpvs"][:, c]) for c in range(N)]
    R, F = np.array(out["reference"]), np.array(out["fast"])
    z = (F.mean(axis=0) - R.mean(axis=0)) / np.sqrt(R.var(axis=0, ddof=1) / R.shape[0] + F.var(axis=0, ddof=1) / F.shape[0])
    assert np.all(np.abs(z) < 4.0), z


def test_baseline_config4_shape_tempered_transdimensional_ladder():
    """BASELINE configs[4] at its own shape, on one GPU: 64 ladders x 8 temperatures (geometric 1..30), up to 20
    layers (transdimensional), Rayleigh + Love phase dispersion + P receiver function.  Invariants (tempering has
    no reference counterpart): beta = 1 everywhere without swaps is the plain run bit for bit; with the ladder every
    ladder keeps its multiset of temperatures through all sweeps; swaps happen; the cold chains sit at a higher
    likelihood than the hot ones; cold-only samples hold one column per ladder."""
    g = golden("chain_golden.npz")
    t1 = bh.RayleighDispersionPhase(g["xsw"], g["ysw"])
    # a Love target on the same periods: synthetic observed data from the engine itself (no reference Love data
    # in the chain fixture); the invariants below do not depend on what is observed
    from bayhunter_amd.synth import true_model
    nlay, h, vp, vs, rho = true_model(6)
    yl, err = bh.default_engine(0).swd_batch(nlay, h, vp, vs, rho, g["xsw"], 1, 0)
    assert err[0] == 0
    t2 = bh.LoveDispersionPhase(g["xsw"], yl[0])
    t3 = bh.PReceiverFunction(g["xrf"], g["yrf"])
    t3.moddata.plugin.set_modelparams(gauss=1.0, p=6.4)
    priors = dict(vpvs=(1.4, 2.1), layers=(1, 20), vs=(2, 5), z=(0, 60), rfnoise_corr=(0.35, 0.75),
                  rfnoise_sigma=(1e-5, 0.05), swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.1))
    init = dict(iter_burnin=500, iter_main=200, acceptance=(40, 45), thickmin=0.1, lvz=None, hvz=None, rcond=None, maxmodels=20)
    nl, nr = 64, 8
    C = nl * nr
    ladder = np.arange(C) % nl
    betas = (1.0 / np.geomspace(1.0, 30.0, nr))[np.arange(C) // nl]

    def targets():
        return bh.JointTarget([t1, t2, t3])

    plain = DeviceChains(targets(), C, init, priors, seed=31).run().state_host()
    ones = DeviceChains(targets(), C, init, priors, seed=31, betas=np.ones(C), ladder=ladder, swap_every=0).run().state_host()
    for k in ("n", "vs", "z", "like", "noise", "vpvs", "propdist", "accepted"):
        assert np.array_equal(plain[k], ones[k]), k
    assert plain["n"].max() > 6 and plain["n"].min() >= 1                     # transdimensional: depths differ
    dc = DeviceChains(targets(), C, init, priors, seed=31, betas=betas, ladder=ladder, swap_every=25)
    while dc.iiter < dc.iter_phase2:
        if dc.iiter % dc.thinning == 0:
            dc._snapshot()
        dc.iterate()
        if dc.iiter % 100 == 0:                                               # every ladder keeps its temperatures
            b = dc.state_host()["beta"]
            for lid in (0, 17, 63):
                assert np.array_equal(np.sort(b[ladder == lid]), np.sort(betas[ladder == lid]))
    st = dc.state_host()
    assert dc.sweep == 700 // 25 and dc.nswaps > 200
    for lid in range(nl):
        assert np.array_equal(np.sort(st["beta"][ladder == lid]), np.sort(betas[ladder == lid]))
    cold, hot = st["beta"] == 1.0, st["beta"] == betas.min()
    assert cold.sum() == nl and hot.sum() == nl
    assert np.median(st["like"][cold]) > np.median(st["like"][hot])
    s = dc.samples("p2", cold_only=True)
    assert s["models"].shape[:2] == (len(dc.snap["p2"]), nl) and (s["beta"] == 1.0).all()
    assert s["models"].shape[2] == 2 * 21


def test_initial_state_and_windows_do_not_depend_on_the_shard_size():
    """the chains' evaluation calls run the trial-per-lane kernel with the trials per round PINNED (DeviceChains.TRIALS)
    -- the windows AND the host-driven initial state -- so a shard's chains get the bits they get in the whole job whatever the
    shard's size.  600 chains x 2 dispersion targets are 1200 (model, target) pairs (32 trials by the call's shape), a shard of
    150 of them 300 pairs (64 by shape): initial likelihoods and the states after a few windows agree bit for bit."""
    g = golden("chain_golden.npz")
    su = SETUPS["exp"]

    def targets():
        t1 = bh.RayleighDispersionPhase(g["xsw"], g["ysw"])
        t2 = bh.LoveDispersionPhase(g["xsw"], 1.05 * g["ysw"])
        return bh.JointTarget([t1, t2])
    priors = dict(su["priors"])
    init = dict(su["init"], iter_burnin=60, iter_main=20, maxmodels=5)
    whole = DeviceChains(targets(), 600, init, priors, seed=11, spec_depth=2)     # (the same window depth: it is chosen by the chain count otherwise)
    shard = DeviceChains(targets(), 150, init, priors, seed=11, chain_offset=300, spec_depth=2)
    a, b = whole.state_host(), shard.state_host()
    sl = slice(300, 450)
    assert np.array_equal(a["like"][sl], b["like"]) and np.array_equal(a["misfits"][:, sl], b["misfits"])
    assert np.array_equal(a["vs"][:, sl], b["vs"], equal_nan=True)
    for _ in range(4):
        whole.iterate()
        shard.iterate()
    assert whole.iiter == shard.iiter
    a, b = whole.state_host(), shard.state_host()
    assert np.array_equal(a["like"][sl], b["like"]) and np.array_equal(a["n"][sl], b["n"])
    assert np.array_equal(a["vs"][:, sl], b["vs"], equal_nan=True) and np.array_equal(a["accepted"][:, sl], b["accepted"])
