"""Markov chains advanced in lock-step on the engine: the caller of the hot path.

Mirror of the reference's per-chain sampler (`SingleChain`, src/SingleChain.py:25-690) and of
the part of `MCMC_Optimizer` that creates the chains (src/mcmcOptimizer.py:31-138), re-shaped for
a GPU: instead of one OS process per chain, each evaluating ONE proposal at a time, all chains of
a rank form a batch -- every iteration each chain draws its proposal on the host exactly as the
reference does (same move types, same prior/validity rules, same `numpy.random.RandomState`
draw order per chain, App. F of SURVEY.md), the valid proposals of all chains go through ONE
`JointTarget.evaluate_batch` call (forward models + likelihood on the device), and each chain then
takes its accept/reject decision.  Chains stay independent, so the per-chain trajectory is the
reference's trajectory: with the same seed it reproduces the reference's accepted-model sequence
(tests/test_gpu_chains.py replays a recorded `SingleChain` run).

Results are written in the reference's format (`c%03d_p{1,2}{models,likes,misfits,noise,vpvs}.npy`,
SingleChain.py:646-690) so that the reference's `PlotFromStorage` keeps working on engine output.
"""
import copy
import os
import os.path as op

import numpy as np

from .Models import Model
from .Targets import JointTarget, select_noise_laws

PAR_MAP = {"vsmod": 0, "zvmod": 1, "birth": 2, "death": 2, "noise": 3, "vpvs": 4}  # SingleChain.py:21-22

# src/defaults/defaults.ini
DEFAULT_PRIORS = dict(mantle=None, vpvs=(1.5, 2.1), layers=(1, 20), vs=(1, 5), z=(0, 60), mohoest=None,
                      rfnoise_corr=(0.35, 0.75), rfnoise_sigma=(1e-5, 0.05), swdnoise_corr=0.,
                      swdnoise_sigma=(1e-5, 0.1))
DEFAULT_INITPARAMS = dict(nchains=3, iter_burnin=2048 * 2, iter_main=2048, propdist=(0.025, 0.025, 0.015, 0.005, 0.005),
                          acceptance=(40, 45), thickmin=0., lvz=None, hvz=None, rcond=None, station="test",
                          savepath="results/", maxmodels=50000)


def _is_fixed(prior):
    """A scalar prior = a fixed parameter.  The reference tests `type(p) in [int, float, np.float64]` for noise
    (src/SingleChain.py:137) and `type(p) == float` for vp/vs (:153, :598); every scalar it does not recognise
    (an int or numpy-scalar vp/vs, a float32 noise value) makes it fail on `p[0]` a few lines later, so each
    configuration the reference RUNS is classified identically here -- the broader test only accepts what the
    reference raises on."""
    return isinstance(prior, (int, float, np.floating)) and not isinstance(prior, bool)


class _Chain(object):
    """State of one chain (the attributes of SingleChain that survive between iterations)."""

    def __init__(self, idx, seed, propdist):
        self.idx = idx
        self.rstate = np.random.RandomState(seed)
        self.propdist = np.array(propdist, dtype=float)
        self.accepted = np.zeros(self.propdist.size)
        self.proposed = np.zeros(self.propdist.size)
        self.n = 0
        self.dvs2 = 0.0
        # accepted rows (stored as float32 like the reference's shared arrays, mcmcOptimizer.py:92-125)
        self.models, self.misfits, self.likes, self.noise, self.vpvs, self.iters = [], [], [], [], [], []


class ChainBatch(object):
    """`nchains` independent rj-McMC chains sharing one set of targets and one engine."""

    def __init__(self, targets, seeds, initparams=None, modelpriors=None, search="reference", arith="exact", trials=None):
        """trials: trials per round of the trial-per-lane kernel in the chains' evaluation calls (Engine.set_swd_trials; None =
        the engine's setting, i.e. by the call's shape unless pinned).  A driver whose results must not depend on how many
        chains an evaluation call holds pins it (DeviceChains: 32, for its windows AND for this initial state).
        search: root refinement of the dispersion search in the chains' evaluation calls (Engine.set_swd_search).  The
        default is the REFERENCE's sequence, whatever the engine's own setting: these chains walk in the reference's order
        from the reference's seeds, and a recorded run of `SingleChain` replays exactly only with the reference's bits.
        arith: the arithmetic where every dispersion target of a call takes the short refinement (Engine.set_swd_arith)."""
        self.search = search
        self.arith = arith
        self.trials = trials
        self.targets = targets if isinstance(targets, JointTarget) else JointTarget(targets)
        self.priors = dict(DEFAULT_PRIORS)
        self.priors.update(modelpriors or {})
        self.initparams = dict(DEFAULT_INITPARAMS)
        self.initparams.update(initparams or {})
        ip, pr = self.initparams, self.priors
        self.dv = pr["vs"][1] - pr["vs"][0]
        self.iter_phase1 = int(ip["iter_burnin"])
        self.iter_phase2 = int(ip["iter_main"])
        self.iterations = self.iter_phase1 + self.iter_phase2
        self.iiter = -self.iter_phase1
        self.acceptance = ip["acceptance"]
        self.thickmin = ip["thickmin"]
        self.maxlayers = int(pr["layers"][1]) + 1
        self.lowvelperc, self.highvelperc, self.mantle = ip["lvz"], ip["hvz"], pr["mantle"]
        self.ntargets = self.targets.ntargets
        self.chains = [_Chain(i, s, ip["propdist"]) for i, s in enumerate(seeds)]
        self._init_models_and_currentvalues()

    # ---- initial state (SingleChain.py:71-157) ----------------------------------------------------
    def _init_models_and_currentvalues(self):
        pr = self.priors
        self.noisepriors = []
        for t in self.targets.targets:
            for ref in ("noise_corr", "noise_sigma"):
                self.noisepriors.append(pr[t.noiseref + ref])
        corrfix = np.array([_is_fixed(p) for p in self.noisepriors])
        self.noiseinds = np.where(~corrfix)[0]
        init = []
        for c in self.chains:
            c.currentvpvs = pr["vpvs"] if _is_fixed(pr["vpvs"]) else c.rstate.uniform(low=pr["vpvs"][0], high=pr["vpvs"][1])
            imodel = self._draw_initmodel(c)
            inoise = np.ones(2 * self.ntargets) * np.nan
            for i, prior in enumerate(self.noisepriors):
                inoise[i] = prior if corrfix[i] else c.rstate.uniform(low=prior[0], high=prior[1])
            init.append((imodel, inoise))
        # covariance laws: identical for every chain (same priors), SingleChain.py:159-205
        select_noise_laws(self.targets.targets, corrfix[::2], init[0][1][::2], self.initparams["rcond"])
        logL, misfits = self._evaluate([(m, n, c.currentvpvs) for (m, n), c in zip(init, self.chains)])
        for c, (m, n), l, mf in zip(self.chains, init, logL, misfits):
            self._accept(c, m, n, c.currentvpvs, l, mf)
            self._append(c)
        self.modelmods = ["vsmod", "zvmod", "birth", "death"]
        self.noisemods = [] if len(self.noiseinds) == 0 else ["noise"]
        self.vpvsmods = [] if _is_fixed(pr["vpvs"]) else ["vpvs"]
        self.modifications = self.modelmods + self.noisemods + self.vpvsmods

    def _draw_initmodel(self, c):
        pr = self.priors
        zmin, zmax = pr["z"]
        vsmin, vsmax = pr["vs"]
        layers = pr["layers"][0] + 1  # half space
        while True:
            vs = c.rstate.uniform(low=vsmin, high=vsmax, size=layers)
            vs.sort()
            if pr["mohoest"] is not None and layers > 1:
                mean, std = pr["mohoest"]
                moho = c.rstate.normal(loc=mean, scale=std)
                tmp_z = c.rstate.uniform(1, np.min([5, moho]))
                tmp = [moho - tmp_z, moho + tmp_z]
                z_vnoi = np.array(tmp) if layers - 2 == 0 else np.concatenate(
                    (tmp, c.rstate.uniform(low=zmin, high=zmax, size=(layers - 2))))
            else:
                z_vnoi = c.rstate.uniform(low=zmin, high=zmax, size=layers)
            z_vnoi.sort()
            model = np.concatenate((vs, z_vnoi))
            if self._validmodel(c, model):
                return model

    # ---- proposals (SingleChain.py:246-328, :394-420) -----------------------------------------------
    def _get_modelproposal(self, c, modify):
        model = copy.copy(c.currentmodel)
        rs = c.rstate
        if modify == "vsmod":
            ind = rs.randint(0, model.size / 2)
            model[ind] = model[ind] + rs.normal(0, c.propdist[0])
        elif modify == "zvmod":
            ind = rs.randint(model.size / 2, model.size)
            model[ind] = model[ind] + rs.normal(0, c.propdist[1])
        elif modify == "birth":
            n, vs_vnoi, z_vnoi = Model.split_modelparams(model)
            z_birth = rs.uniform(low=self.priors["z"][0], high=self.priors["z"][1])
            ind = np.argmin(abs(z_vnoi - z_birth))  # closest nucleus
            vs_before = vs_vnoi[ind]
            vs_birth = vs_before + rs.normal(0, c.propdist[2])
            c.dvs2 = np.square(vs_birth - vs_before)
            model = np.concatenate((vs_vnoi, [vs_birth], z_vnoi, [z_birth]))
        elif modify == "death":
            n, vs_vnoi, z_vnoi = Model.split_modelparams(model)
            ind_death = rs.randint(low=0, high=z_vnoi.size)
            z_before, vs_before = z_vnoi[ind_death], vs_vnoi[ind_death]
            z_new, vs_new = np.delete(z_vnoi, ind_death), np.delete(vs_vnoi, ind_death)
            ind = np.argmin(abs(z_new - z_before))
            c.dvs2 = np.square(vs_new[ind] - vs_before)
            model = np.concatenate((vs_new, z_new))
        n, vs, z_vnoi = Model.split_modelparams(model)  # re-sort by nucleus depth if needed (:315-328)
        if not np.all(np.diff(z_vnoi) > 0):
            ind = np.argsort(z_vnoi)
            model = np.concatenate((vs[ind], z_vnoi[ind]))
        return model

    def _validmodel(self, c, model):
        """Prior and validity rules, SingleChain.py:330-392 (uses the chain's CURRENT vp/vs)."""
        pr = self.priors
        vp, vs, h = Model.get_vp_vs_h(model, c.currentvpvs, self.mantle)
        layermodel = h.size - 1
        if not (pr["layers"][0] <= layermodel <= pr["layers"][1]):
            return False
        if np.any(h[:-1] < self.thickmin):
            return False
        if np.any(vs < pr["vs"][0]) or np.any(vs > pr["vs"][1]):
            return False
        z = np.cumsum(h)
        if np.any(z < pr["z"][0]) or np.any(z > pr["z"][1]):
            return False
        if self.lowvelperc is not None:
            compvels = vs[1:] - (vs[:-1] * (1 - self.lowvelperc))
            if not compvels.size == compvels[compvels > 0].size:
                return False
        if self.highvelperc is not None:
            compvels = (vs[:-1] * (1 + self.highvelperc)) - vs[1:]
            if not compvels.size == compvels[compvels > 0].size:
                return False
        return True

    def _propose(self, c):
        """One chain's proposal for this iteration -> (modify, model, noise, vpvs) or None."""
        if self.iiter < (-self.iter_phase1 + (self.iterations * 0.01)):
            modify = c.rstate.choice(["vsmod", "zvmod"] + self.noisemods + self.vpvsmods)  # no birth/death yet
        else:
            modify = c.rstate.choice(self.modifications)
        if modify in self.modelmods:
            model = self._get_modelproposal(c, modify)
            return (modify, model, c.currentnoise, c.currentvpvs) if self._validmodel(c, model) else None
        if modify in self.noisemods:
            noise = copy.copy(c.currentnoise)
            ind = c.rstate.choice(self.noiseinds)
            noise[ind] = noise[ind] + c.rstate.normal(0, c.propdist[3])
            for idx in self.noiseinds:
                if noise[idx] < self.noisepriors[idx][0] or noise[idx] > self.noisepriors[idx][1]:
                    return None
            return (modify, c.currentmodel, noise, c.currentvpvs)
        vpvs = c.currentvpvs + c.rstate.normal(0, c.propdist[4])
        if vpvs < self.priors["vpvs"][0] or vpvs > self.priors["vpvs"][1]:
            return None
        return (modify, c.currentmodel, c.currentnoise, vpvs)

    # ---- the hot path: all valid proposals in one device call ----------------------------------------
    def _evaluate(self, proposals):
        """proposals: list of (model, noise, vpvs) -> (logL list, misfits list)."""
        B = len(proposals)
        if B == 0:
            return [], []
        Lmax = self.maxlayers
        nlay = np.zeros(B, dtype=np.int32)
        h = np.zeros((Lmax, B)); vp = np.zeros((Lmax, B)); vs = np.zeros((Lmax, B))
        noise = np.zeros((B, 2 * self.ntargets))
        for b, (model, nz, vpvs) in enumerate(proposals):
            pvp, pvs, ph = Model.get_vp_vs_h(model, vpvs, self.mantle)
            n = pvs.size
            nlay[b] = n
            h[:n, b], vp[:n, b], vs[:n, b] = ph, pvp, pvs
            noise[b] = nz
        with self.targets.engine.searching(self.search), self.targets.engine.computing(self.arith), self.targets.engine.trying(self.trials):
            logL, misfits, err = self.targets.evaluate_batch(nlay, h, vp, vs, noise)
        return list(logL), [m for m in misfits]

    # ---- accept / store (SingleChain.py:452-509) ------------------------------------------------------
    def _acceptance_probability(self, c, modify, proplike):
        if modify in ("vsmod", "zvmod", "noise", "vpvs"):
            return proplike - c.currentlikelihood
        theta = c.propdist[2]
        if modify == "birth":
            A = (theta * np.sqrt(2 * np.pi)) / self.dv
            return np.log(A) + c.dvs2 / (2. * np.square(theta)) + (proplike - c.currentlikelihood)
        A = self.dv / (theta * np.sqrt(2 * np.pi))
        return np.log(A) - c.dvs2 / (2. * np.square(theta)) + (proplike - c.currentlikelihood)

    def _accept(self, c, model, noise, vpvs, like, misfits):
        c.currentmisfits, c.currentlikelihood = misfits, like
        c.currentmodel, c.currentnoise, c.currentvpvs = model, noise, vpvs
        c.lastmoditer = self.iiter

    def _append(self, c):
        row = np.full(self.maxlayers * 2, np.nan, dtype=np.float32)
        row[:c.currentmodel.size] = c.currentmodel
        c.models.append(row)
        c.misfits.append(np.asarray(c.currentmisfits, dtype=np.float32))
        c.likes.append(np.float32(c.currentlikelihood))
        c.noise.append(np.asarray(c.currentnoise, dtype=np.float32))
        c.vpvs.append(np.float32(c.currentvpvs))
        c.iters.append(self.iiter)
        c.n += 1

    def _adjust_propdist(self, c):
        """SingleChain.py:425-450"""
        with np.errstate(invalid="ignore", divide="ignore"):
            rate = c.accepted / c.proposed * 100
        for i, r in enumerate(rate):
            if np.isnan(r):
                continue
            if r < self.acceptance[0]:
                c.propdist[i] = max(c.propdist[i] * 0.95, 0.001)
            elif r > self.acceptance[1]:
                c.propdist[i] = c.propdist[i] * 1.05

    # ---- one lock-step iteration of every chain (SingleChain.py:511-589) -----------------------------
    def iterate(self):
        props = [self._propose(c) for c in self.chains]
        live = [i for i, p in enumerate(props) if p is not None]
        logL, misfits = self._evaluate([props[i][1:] for i in live])
        for i, like, mf in zip(live, logL, misfits):
            c = self.chains[i]
            modify, model, noise, vpvs = props[i]
            paridx = PAR_MAP[modify]
            c.proposed[paridx] += 1
            u = np.log(c.rstate.uniform(0, 1))
            alpha = self._acceptance_probability(c, modify, like)
            if u < alpha:
                self._accept(c, model, noise, vpvs, like, mf)
                self._append(c)
                c.accepted[paridx] += 1
            if self.iiter % 1000 == 0 and np.all(c.proposed) != 0:
                self._adjust_propdist(c)
        self.iiter += 1

    def run(self, progress=None):
        self.iiter = -self.iter_phase1
        while self.iiter < self.iter_phase2:
            self.iterate()
            if progress is not None and self.iiter % 1000 == 0:
                progress(self)
        return self

    # ---- results in the reference's format (SingleChain.py:591-690, Models.py:227-274) ---------------
    def chain_arrays(self, c):
        c = self.chains[c] if isinstance(c, int) else c
        return dict(models=np.array(c.models, dtype=np.float32).reshape(c.n, self.maxlayers * 2),
                    misfits=np.array(c.misfits, dtype=np.float32).reshape(c.n, self.ntargets + 1),
                    likes=np.array(c.likes, dtype=np.float32), noise=np.array(c.noise, dtype=np.float32).reshape(c.n, -1),
                    vpvs=np.array(c.vpvs, dtype=np.float32), iters=np.array(c.iters, dtype=float))

    def save(self, savepath=None):
        """Write c%03d_p{1,2}{models,likes,misfits,noise,vpvs}.npy, rows repeated by the number of
        iterations the model stayed current and thinned to `maxmodels`."""
        savepath = op.join(savepath or self.initparams["savepath"], "data")
        os.makedirs(savepath, exist_ok=True)
        written = []
        for c in self.chains:
            a = self.chain_arrays(c)
            p2 = a["iters"] >= 0
            nmain = None
            for tag, sel, final in (("p1", ~p2, 0), ("p2", p2, self.iiter)):
                if not np.any(sel):
                    continue
                w = np.diff(np.concatenate((a["iters"][sel], [final]))).astype(int)
                rep = {k: np.repeat(a[k][sel], w, axis=0) for k in ("models", "likes", "misfits", "noise", "vpvs")}
                if tag == "p2":
                    nmain = rep["likes"].size
                thinning = int(np.ceil(float(nmain if nmain else rep["likes"].size) / float(self.initparams["maxmodels"])))
                rep["_thin"] = max(1, thinning)
                written.append((c.idx, tag, rep))
        # the reference computes ONE thinning factor per chain from the main phase (:639-641)
        thin = {}
        for idx, tag, rep in written:
            if tag == "p2":
                thin[idx] = rep["_thin"]
        for idx, tag, rep in written:
            t = thin.get(idx, rep["_thin"])
            for k in ("models", "likes", "misfits", "noise", "vpvs"):
                np.save(op.join(savepath, "c%.3d_%s%s" % (idx, tag, k)), rep[k][::t])
        from .results import save_config
        save_config(self.targets, op.join(savepath, "%s_config.pkl" % self.initparams.get("station", "test")),
                    priors=self.priors, initparams=self.initparams)
        return savepath


class MCMC_Optimizer(object):
    """Drop-in for the reference's optimizer front-end (src/mcmcOptimizer.py:31-138, :202-258):
    same constructor; `mp_inversion` runs all chains as one batch on the GPU instead of forking one
    process per chain.  Chain seeds are drawn as the reference draws them (`rstate.randint(1000)`)."""

    def __init__(self, targets, initparams=None, priors=None, random_seed=None):
        self.rstate = np.random.RandomState(random_seed)
        self.initparams = dict(DEFAULT_INITPARAMS)
        self.initparams.update(initparams or {})
        self.priors = dict(DEFAULT_PRIORS)
        self.priors.update(priors or {})
        self.nchains = self.initparams["nchains"]
        seeds = [self.rstate.randint(1000) for _ in range(self.nchains)]
        self.batch = ChainBatch(targets, seeds, self.initparams, self.priors)

    def mp_inversion(self, baywatch=False, dtsend=0.5, nthreads=0):
        """All chains as one lock-step batch; writes the reference's result folder (per-chain files + the
        configuration pickle `PlotFromStorage` opens).  `nthreads` / `dtsend` have no meaning here (no process
        farm); BayWatch live streaming is not part of this package."""
        if baywatch:
            import warnings
            warnings.warn("baywatch=True: live streaming to BayWatch is not provided by bayhunter_amd (ignored)")
        self.batch.run()
        return self.batch.save()
