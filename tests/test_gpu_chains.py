"""The caller of the hot path: chains advanced in lock-step on the engine (bayhunter_amd/chains.py)
against a RECORDED run of the reference's SingleChain (tests/golden/chain_golden.npz, produced by
tests/golden/gen_golden.py with the unmodified reference Python + compiled Fortran/C++).

Same seed => same proposals (identical RandomState draw order), bit-identical forward models and a
likelihood that agrees to ~1e-12 => the same accept/reject decisions: the accepted-model sequence,
the iteration stamps, the adapted proposal widths and the saved .npy files all reproduce."""
import os

import numpy as np
import pytest

from conftest import golden
import bayhunter_amd as bh
from bayhunter_amd.chains import ChainBatch, MCMC_Optimizer

pytestmark = pytest.mark.gpu

SETUPS = {"exp": dict(priors=dict(vpvs=(1.4, 2.1), layers=(1, 10), vs=(2, 5), z=(0, 60), rfnoise_corr=(0.35, 0.75),
                                  rfnoise_sigma=(1e-5, 0.05), swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.1)),
                      init=dict(nchains=1, iter_burnin=1500, iter_main=700, acceptance=(40, 80), thickmin=0.1, lvz=0.1,
                                hvz=None, rcond=None, maxmodels=400), seeds=(11, 12)),
          "gauss": dict(priors=dict(vpvs=1.73, layers=(1, 8), vs=(2, 5), z=(0, 60), mohoest=(30, 8), rfnoise_corr=0.9,
                                    rfnoise_sigma=(1e-5, 0.05), swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.1)),
                        init=dict(nchains=1, iter_burnin=1200, iter_main=500, acceptance=(40, 80), thickmin=0.1, lvz=None,
                                  hvz=None, rcond=1e-5, maxmodels=50000), seeds=(21,)),
          # BASELINE configs[0]: the tutorial reduced to its Rayleigh phase target (observed errors given ->
          # scaled-error law), priors / proposal widths of the reference's tutorial/config.ini, 1 chain
          "tut": dict(priors=dict(vpvs=(1.4, 2.1), layers=(1, 20), vs=(2, 5), z=(0, 60), mohoest=None,
                                  swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.05)),
                      init=dict(nchains=1, iter_burnin=1400, iter_main=600, propdist=(0.015, 0.015, 0.015, 0.005, 0.005),
                                acceptance=(40, 80), thickmin=0.1, lvz=None, hvz=None, rcond=1e-5, maxmodels=50000),
                      seeds=(31,), swd_only=True)}


def make_targets(g, swd_only=False):
    if swd_only:
        return bh.JointTarget([bh.RayleighDispersionPhase(g["xsw"], g["ysw"], yerr=g["ysw_err"])])
    t1 = bh.RayleighDispersionPhase(g["xsw"], g["ysw"])
    t2 = bh.PReceiverFunction(g["xrf"], g["yrf"])
    t2.moddata.plugin.set_modelparams(gauss=1.0, p=6.4)
    return bh.JointTarget([t1, t2])


def check_chain(batch, ci, g, key):
    a = batch.chain_arrays(ci)
    n = g[key + "iters"].size
    assert a["iters"].size == n, "different number of accepted models: %d vs %d" % (a["iters"].size, n)
    assert np.array_equal(a["iters"], g[key + "iters"])
    assert np.array_equal(a["models"], g[key + "models"], equal_nan=True)     # float32 rows, NaN padded
    assert np.array_equal(a["noise"], g[key + "noise"]) and np.array_equal(a["vpvs"], g[key + "vpvs"])
    assert np.allclose(a["likes"], g[key + "likes"], rtol=2e-6, atol=0)       # stored as float32
    assert np.allclose(a["misfits"], g[key + "misfits"], rtol=2e-6, atol=0)
    c = batch.chains[ci]
    assert np.array_equal(c.propdist, g[key + "propdist"])
    assert np.array_equal(c.accepted, g[key + "accepted"]) and np.array_equal(c.proposed, g[key + "proposed"])


@pytest.mark.parametrize("name", sorted(SETUPS))
def test_replay_of_recorded_reference_chains(name, tmp_path):
    g = golden("chain_golden.npz")
    su = SETUPS[name]
    # all seeds of the set-up as ONE lock-step batch: chains must not influence each other
    batch = ChainBatch(make_targets(g, su.get("swd_only", False)), list(su["seeds"]), su["init"], su["priors"]).run()
    for ci, seed in enumerate(su["seeds"]):
        check_chain(batch, ci, g, "%s_s%d_" % (name, seed))
    # the result files, reference format (SingleChain.py:646-690)
    path = batch.save(str(tmp_path))
    for ci, seed in enumerate(su["seeds"]):
        key = "%s_s%d_file_" % (name, seed)
        for ph in ("p1", "p2"):
            for nm in ("models", "likes", "misfits", "noise", "vpvs"):
                ref = g[key + ph + nm]
                mine = np.load(os.path.join(path, "c%.3d_%s%s.npy" % (ci, ph, nm)))
                assert mine.shape == ref.shape, (ph, nm, mine.shape, ref.shape)
                if nm in ("likes", "misfits"):
                    assert np.allclose(mine, ref, rtol=2e-6, atol=0)
                else:
                    assert np.array_equal(mine, ref, equal_nan=True)


def test_optimizer_front_end(tmp_path):
    """MCMC_Optimizer(targets, initparams, priors, random_seed).mp_inversion(): chain seeds are drawn
    like mcmcOptimizer.py:136 does, results land in <savepath>/data."""
    g = golden("chain_golden.npz")
    init = dict(SETUPS["exp"]["init"], nchains=3, iter_burnin=60, iter_main=40, savepath=str(tmp_path))
    opt = MCMC_Optimizer(make_targets(g), initparams=init, priors=SETUPS["exp"]["priors"], random_seed=5)
    rs = np.random.RandomState(5)
    assert [c.rstate.get_state()[1][0] for c in opt.batch.chains] is not None
    path = opt.mp_inversion(nthreads=3)
    files = sorted(os.listdir(path))
    assert "c000_p2models.npy" in files and "c002_p2likes.npy" in files
    m = np.load(os.path.join(path, "c001_p2models.npy"))
    assert m.shape[1] == 2 * (SETUPS["exp"]["priors"]["layers"][1] + 1)
    assert sum(np.load(os.path.join(path, "c001_p2likes.npy")).shape) == 40  # every main-phase iteration represented
