"""Device-resident chains: thousands of rj-McMC chains advanced in lock-step with no host work
per chain.

One iteration = three enqueues on the engine's stream (include/bh_engine.h):
    bh_chain_propose  ->  bh_evaluate_batch (device pointers)  ->  bh_chain_accept
so the per-chain Python of the reference's sampler (src/SingleChain.py:511-589 `iterate`, and the
per-chain loops of `bayhunter_amd.chains.ChainBatch`) disappears from the loop; the host only
counts iterations and takes thinned snapshots of the chain states.

Few chains (BASELINE configs[3]: 8 per GPU) leave the GPU idle: one evaluation launch costs ~1.5 ms
whether it holds 8 or 500 models (the length of ONE model's dispersion root search).  `spec_depth` = d
advances every chain by d iterations per launch instead: the proposals of both outcomes of each of the
next d accept/reject decisions are written down first (bh_chain_propose_window: a binary tree of
2^d - 1 proposals per chain; the draws are a pure function of (chain, iteration)), all of them are
evaluated in ONE bh_evaluate_batch, and bh_chain_accept_window walks the realised path -- the
trajectory of the sequential walk, bit for bit.  Windows end where something outside a chain's own
state changes: proposal-width adaptation (every 1000th iteration), snapshots, temperature exchanges.

Random numbers are counter-based (Philox4x32-10) on the device, so a run is reproducible from
`seed` but is NOT the reference's Mersenne-Twister trajectory: `ChainBatch` is the draw-for-draw
replay of the reference, this class is the throughput mode.  The proposal / validity / acceptance
arithmetic is the same and is tested against an independent numpy restatement with injected draws
(tests/test_gpu_device_chains.py).

The initial state (initial models, noise, covariance-law selection, first likelihood) is produced by
`ChainBatch` exactly as the reference does (SingleChain.py:71-205) and uploaded once.

Storage: the reference keeps every accepted model with its iteration stamp and, when saving,
repeats each by its dwell time and thins to `maxmodels` rows (SingleChain.py:591-690) -- i.e. it
keeps the chain's current model at every `thinning`-th iteration.  Here that sampling is done on
the fly: every `thinning`-th iteration the current state of all chains is copied out, which is the
same estimator without storing what would be thinned away.  (One difference: the reference's
main-phase file starts with the first model ACCEPTED at iteration >= 0; here the main phase starts
with the model that is current at iteration 0.)
"""
import ctypes as C
import os
import os.path as op

import numpy as np

from .chains import ChainBatch, DEFAULT_INITPARAMS, DEFAULT_PRIORS, _is_fixed
from .engine import BH_CHAIN_MAXDEPTH, BH_CHAIN_MAXLAYERS, ChainConfig, ChainState, EngineError
from .Targets import JointTarget


def auto_spec_depth(nchains, budget=None):
    """Speculation depth for `nchains` chains on one GPU: the deepest tree whose nodes (chains x (2^d - 1)
    evaluations per launch) stay within `budget` evaluations -- below ~1000 models a launch of the dispersion kernel
    costs about what 8 models cost (docs/HISTORY.md 3.1: 1.5 ms at B <= 512, 2.0 ms at 1024, 3.0 ms at 2048), so d
    iterations per launch are nearly free; beyond it the launch time grows faster than the depth.
    BH_SPEC_BUDGET overrides the budget (0 = no speculation)."""
    if budget is None:
        budget = int(os.environ.get("BH_SPEC_BUDGET", "1024"))
    d = 1
    while d < BH_CHAIN_MAXDEPTH and nchains * ((1 << (d + 1)) - 1) <= budget:
        d += 1
    return d


class DeviceChains(object):
    TRIALS = 32  # trials per round of the trial-per-lane kernel in every evaluation call of the chains (windows and initial state)

    def __init__(self, targets, nchains, initparams=None, modelpriors=None, seed=0, device=None, inject=False,
                 betas=None, ladder=None, swap_every=0, dist=None, chain_offset=None, spec_depth=None, search="fast", arith="fast"):
        """`nchains` chains on THIS rank.  Sharded jobs (one process per GPU, `dist` = an initialised
        torch.distributed): `seed` is the JOB's seed, the same on every rank; the chains are numbered globally
        (`chain_offset` = global index of this rank's first chain, default: ranks own consecutive blocks in rank
        order) and every chain's random stream and initial state depend on (seed, global index) only -- N ranks x C
        chains walk exactly the trajectories of one rank with N*C chains.
        betas / ladder / swap_every: parallel tempering (no counterpart in the reference).  `betas[c]` is the
        inverse temperature chain c starts with, `ladder[c]` the id of the temperature ladder it belongs to (ids
        are global across ranks); every `swap_every` iterations neighbouring temperatures of each ladder are
        exchanged (`parallel.tempering_exchange`, decisions drawn from the job seed: identical on every rank;
        chains keep their states and swap betas, so nothing but (logL, beta, ladder) of each chain crosses GPUs).
        Posterior samples are the snapshots of the chains that hold beta = 1 at that time
        (`samples(cold_only=True)`, and what `save()` writes).
        device: CUDA device index; default = the engine's (`JointTarget(..., engine=)`), else 0.
        spec_depth: iterations per evaluation launch (speculative window, 1..7; module docstring).  None = chosen
        from the number of chains so that a launch stays in the latency regime (`auto_spec_depth`); 1 = one
        iteration per launch.  Results do not depend on it.
        search: root refinement of the dispersion search in the chains' evaluation launches (Engine.set_swd_search;
        applied around every launch, the engine's own setting is left as it was).  Default "fast" (the engine's own
        default): fundamental-mode phase velocities within 1.2e-6 relative of the reference's, the reference's failure
        flags (a model the guard fires on -- 2 % of a sampler's Love proposals -- starts again with the reference's
        sequence inside its own wavefront of the window's launch), group velocities the reference's bits -- what the
        chains sample does not change (tests/test_gpu_device_chains.py::test_search_modes_sample_the_same_posterior),
        a window takes 22-27 % less.  "fast_rayleigh": the short refinement for the Rayleigh targets only.
        "reference": the reference's bits throughout (what `ChainBatch`, the replay of recorded reference runs, uses);
        None: whatever the engine is set to.
        arith: arithmetic of those launches where every dispersion target takes the short refinement (Engine.set_swd_arith,
        applied like `search`).  Default "fast" (the engine's own): the windows run the trial-per-lane kernel with 32 trials
        per round whatever their size (Engine.set_swd_trials), so that windows of any depth and shards of any size walk the
        same trajectory; its guarded models (2 % of a sampler's Love proposals) are re-run by a second launch.  "exact": the
        reference's rounding points -- the windows then take the layer-parallel kernel, which restarts a guarded model in
        place.  Measured on MI355X (chain-iterations/s, "fast" / "exact"): 8 chains 7.4e4 / 5.2e4, a 64-chain tempered rung
        2.18e5 / 1.83e5, 512 chains 4.6e5 / 3.9e5.  None: whatever the engine is set to."""
        import torch
        self.torch = torch
        self.targets = targets if isinstance(targets, JointTarget) else JointTarget(targets)
        if self.targets._engine is None and device is not None:
            from .engine import default_engine
            self.targets._engine = default_engine(int(device))   # kernels and tensors on the same GPU
        self.engine = self.targets.engine
        if search not in (None, "reference", "fast", "fast_rayleigh"):
            raise ValueError("search must be None, 'reference', 'fast' or 'fast_rayleigh'")
        self.search = search
        if arith not in (None, "exact", "fast"):
            raise ValueError("arith must be None, 'exact' or 'fast'")
        self.arith = arith
        if device is None:
            device = self.engine.device
        if int(device) != int(self.engine.device):
            raise EngineError("DeviceChains(device=%d) but the targets' engine runs on GPU %d" % (device, self.engine.device))
        self.priors = dict(DEFAULT_PRIORS)
        self.priors.update(modelpriors or {})
        self.initparams = dict(DEFAULT_INITPARAMS)
        self.initparams.update(initparams or {})
        ip, pr = self.initparams, self.priors
        self.C = int(nchains)
        self.nt = self.targets.ntargets
        self.ML = int(pr["layers"][1]) + 1
        if self.ML > BH_CHAIN_MAXLAYERS:
            raise EngineError("priors['layers'][1] + 1 = %d exceeds BH_CHAIN_MAXLAYERS = %d" % (self.ML, BH_CHAIN_MAXLAYERS))
        self.iter_phase1, self.iter_phase2 = int(ip["iter_burnin"]), int(ip["iter_main"])
        self.iterations = self.iter_phase1 + self.iter_phase2
        self.iiter = -self.iter_phase1
        self.thinning = max(1, int(np.ceil(float(self.iter_phase2) / float(ip["maxmodels"]))))
        self.swap_every, self.dist, self._nswaps_host, self.sweep, self.seed = int(swap_every), dist, 0, 0, int(seed)
        self.depth = auto_spec_depth(self.C) if spec_depth is None else int(spec_depth)
        if not 1 <= self.depth <= BH_CHAIN_MAXDEPTH:
            raise EngineError("spec_depth must be 1..%d" % BH_CHAIN_MAXDEPTH)
        self.ld = self.C * ((1 << self.depth) - 1)        # columns of the proposal arrays: all nodes of all chains
        self.snap_in_run = False                          # run(): windows also end at snapshot iterations
        self.launches = 0
        self._hint = 0
        self._dev_exchange = None
        from .parallel import chain_layout, chain_seeds, rank_chain_counts
        self.rank_counts = rank_chain_counts(self.C, dist, int(device))       # collective buffers on THIS rank's GPU
        off, tot = chain_layout(self.C, dist, int(device))
        if chain_offset is not None:
            off = int(chain_offset)
        self.chain_offset, self.C_global = off, max(tot, off + self.C)
        self.rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0

        # ---- initial state through the reference-order host code --------------------------------
        host = ChainBatch(self.targets, chain_seeds(seed, off, self.C), ip, pr,
                          search=self.search if self.search is not None else self.targets.engine.swd_search(),
                          arith=self.arith if self.arith is not None else self.targets.engine.swd_arith(),
                          trials=self.TRIALS)   # (the windows' count: the initial likelihoods are the windows' bits)
        self.noisepriors = host.noisepriors
        self.targets._register()  # constant target data + laws live on the device from here on

        cfg = ChainConfig()
        cfg.nt, cfg.maxlayers = self.nt, self.ML
        cfg.layermin, cfg.layermax = int(pr["layers"][0]), int(pr["layers"][1])
        cfg.iter_burnin, cfg.iterations = self.iter_phase1, self.iterations
        (cfg.vsmin, cfg.vsmax), (cfg.zmin, cfg.zmax) = pr["vs"], pr["z"]
        cfg.thickmin = ip["thickmin"]
        cfg.lvz = -1.0 if ip["lvz"] is None else ip["lvz"]
        cfg.hvz = -1.0 if ip["hvz"] is None else ip["hvz"]
        if _is_fixed(pr["vpvs"]):
            cfg.vpvsmin = cfg.vpvsmax = float(pr["vpvs"])
        else:
            cfg.vpvsmin, cfg.vpvsmax = pr["vpvs"]
        if pr["mantle"] is None:
            cfg.mantle_vs, cfg.mantle_vpvs = -1.0, 0.0
        else:
            cfg.mantle_vs, cfg.mantle_vpvs = pr["mantle"]
        cfg.acc_lo, cfg.acc_hi = ip["acceptance"]
        for i, p in enumerate(self.noisepriors):
            if _is_fixed(p):
                cfg.noise_lo[i] = cfg.noise_hi[i] = float(p)
            else:
                cfg.noise_lo[i], cfg.noise_hi[i] = p
        cfg.seed = int(seed) & (2 ** 64 - 1)
        cfg.chain_offset = off
        self.cfg = cfg

        dev = torch.device("cuda", int(device))
        self.dev = dev
        Cn, ML, nt = self.C, self.ML, self.nt
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        t = {}
        vs0 = np.zeros((ML, Cn)); z0 = np.zeros((ML, Cn)); n0 = np.zeros(Cn, dtype=np.int32)
        noise0 = np.zeros((2 * nt, Cn)); mis0 = np.zeros((nt + 1, Cn))
        like0 = np.zeros(Cn); vpvs0 = np.zeros(Cn); pd0 = np.zeros((5, Cn))
        for c, ch in enumerate(host.chains):
            m = np.asarray(ch.currentmodel, dtype=float)
            n = m.size // 2
            n0[c] = n
            vs0[:n, c], z0[:n, c] = m[:n], m[n:]
            noise0[:, c] = ch.currentnoise
            mis0[:, c] = ch.currentmisfits
            like0[c], vpvs0[c] = ch.currentlikelihood, ch.currentvpvs
            pd0[:, c] = ch.propdist
        t["n"] = torch.from_numpy(n0).to(dev)
        t["vs"], t["z"] = torch.from_numpy(vs0).to(dev), torch.from_numpy(z0).to(dev)
        t["vpvs"], t["noise"] = torch.from_numpy(vpvs0).to(dev), torch.from_numpy(noise0).to(dev)
        t["like"], t["misfits"] = torch.from_numpy(like0).to(dev), torch.from_numpy(mis0).to(dev)
        t["propdist"] = torch.from_numpy(pd0).to(dev)
        t["proposed"], t["accepted"] = torch.zeros((5, Cn), **f64), torch.zeros((5, Cn), **f64)
        t["naccepted"] = torch.zeros(Cn, dtype=torch.int64, device=dev)
        t["beta"] = None if betas is None else torch.as_tensor(np.asarray(betas, dtype=np.float64)).to(dev)
        self.ladder = None if betas is None else np.asarray(ladder if ladder is not None else np.zeros(Cn), dtype=np.int64)
        ld = self.ld                                      # node j of chain c in column j*C + c
        for k in ("pn", "move", "valid", "lay_n"):
            t[k] = torch.zeros(ld, **i32)
        for k in ("pvs", "pz", "lay_h", "lay_vp", "lay_vs", "lay_rho"):
            t[k] = torch.zeros((ML, ld), **f64)
        t["pvpvs"], t["dvs2"] = torch.zeros(ld, **f64), torch.zeros(ld, **f64)
        t["pnoise"] = torch.zeros((ld, 2 * nt), **f64)
        t["inject"] = torch.zeros((self.depth, 6, Cn), **f64) if inject else None
        self.t = t
        # outputs of the evaluate call of the current window
        self.logL = torch.zeros(ld, **f64)
        self.mis = torch.zeros((ld, nt + 1), **f64)
        self.err = torch.zeros(ld, **i32)
        st = ChainState()
        for k in ChainState._fields_:
            v = t[k[0]]
            setattr(st, k[0], None if v is None else v.data_ptr())
        self.state = st
        torch.cuda.synchronize(dev)
        self._ext_stream = torch.cuda.ExternalStream(int(self.engine.stream), device=dev)   # the engine's stream, for torch work
        # replica exchange on the device (one rank, or RCCL): the ladder of every chain of the job is static
        if betas is not None and self.swap_every > 0:
            on_gpu = dist is None or not dist.is_initialized() or dist.get_world_size() == 1 or dist.get_backend() == "nccl"
            if on_gpu and os.environ.get("BH_PT_HOST_EXCHANGE", "0") != "1":
                from .parallel import DeviceExchange, gather_chain_axis
                ladder_all = gather_chain_axis(self.ladder, 0, dist, int(device))
                start = int(sum(self.rank_counts[:self.rank]))         # position of this rank's block in gather order
                mine = slice(start, start + Cn)
                self._dev_exchange = DeviceExchange(ladder_all, self.seed, mine, dev, self.rank_counts)
        self.snap = {"p1": [], "p2": []}

    def window(self):
        """Iterations the next launch may cover: the speculation depth, cut where something outside a chain's own
        state changes -- the proposal-width adaptation (an iteration with iiter % 1000 == 0 is the last of its
        window), a temperature exchange, a snapshot of run(), the end of the run."""
        i = self.iiter
        w = min(self.depth, self.iter_phase2 - i, (-i) % 1000 + 1)
        if self.swap_every > 0 and self.t["beta"] is not None:
            w = min(w, self.swap_every - i % self.swap_every)
        if self.snap_in_run:
            w = min(w, self.thinning - i % self.thinning)
        return max(1, w)

    # ---- one lock-step window of all chains: three enqueues, no synchronisation ---------------------
    HINT_EVERY = 64

    def iterate(self):
        """Advance every chain by `window()` iterations (1 with spec_depth = 1); returns that number."""
        e, t, Cn = self.engine, self.t, self.C
        # transdimensional chains: the dispersion kernel's lane groups and LDS rows are sized for the models the chains
        # hold NOW (typically 5-7 layers in arrays of 21), refreshed every HINT_EVERY launches (one small read-back; run()
        # also does it at every snapshot).  The engine uses it for batches of more than a wavefront's worth of models per
        # SIMD pair (many chains); a window of ~1000 models gets one wavefront per model whatever its depth.
        if self.launches % self.HINT_EVERY == 0:
            # (read on the ENGINE's stream: the accept kernels that write t["n"] run there, not on torch's current stream)
            with self.torch.cuda.stream(self._ext_stream):
                self._hint = int(self.torch.ceil(t["n"].double().mean()).item())
        w = self.window()
        B = Cn * ((1 << w) - 1)
        e.chain_propose_window(self.cfg, self.state, Cn, self.iiter, w, self.ld)
        e.set_typical_layers(self._hint)       # (for this call only: the engine is shared with other callers)
        prev = e.swd_search() if self.search is not None else None
        if prev is not None and prev != self.search:
            e.set_swd_search(self.search)
        prev_arith, prev_trials = (e.swd_arith() if self.arith is not None else None), e.swd_trials()
        if prev_arith is not None and prev_arith != self.arith:
            e.set_swd_arith(self.arith)
        e.set_swd_trials(self.TRIALS)
        try:
            e.evaluate_batch_dev(B, self.ML, t["lay_n"].data_ptr(), t["lay_h"].data_ptr(), t["lay_vp"].data_ptr(),
                                 t["lay_vs"].data_ptr(), t["lay_rho"].data_ptr(), self.ld, 1, t["pnoise"].data_ptr(), self.logL.data_ptr(),
                                 self.mis.data_ptr(), self.err.data_ptr())
        finally:
            e.set_typical_layers(0)
            if prev is not None and prev != self.search:
                e.set_swd_search(prev)
            if prev_arith is not None and prev_arith != self.arith:
                e.set_swd_arith(prev_arith)
            e.set_swd_trials(prev_trials)
        e.chain_accept_window(self.cfg, self.state, Cn, self.iiter, w, self.ld, self.logL.data_ptr(), self.mis.data_ptr())
        self.iiter += w
        self.launches += 1
        if self.swap_every > 0 and t["beta"] is not None and self.iiter % self.swap_every == 0:
            self.exchange()
        return w

    def exchange(self):
        """One replica-exchange sweep (the only step of a sharded job with a collective).  On the GPU (one rank, or
        RCCL) it is enqueued on the engine's stream like an iteration (`parallel.DeviceExchange`); with a CPU
        process group (gloo) the gathered values go through the host (`parallel.tempering_exchange`)."""
        if self._dev_exchange is not None:
            with self.torch.cuda.stream(self._ext_stream):
                self._dev_exchange.sweep(self.t["like"], self.t["beta"], self.sweep, self.dist)
            self.sweep += 1
            return
        from .parallel import tempering_exchange
        self.engine.synchronize()
        newb, nacc = tempering_exchange(self.t["like"], self.t["beta"], self.ladder, self.sweep, self.seed, self.dist)
        self.t["beta"].copy_(newb)
        self.torch.cuda.synchronize(self.dev)
        self.sweep += 1
        self._nswaps_host += nacc

    @property
    def nswaps(self):
        """accepted swaps so far (all ranks)"""
        return self._nswaps_host + (int(self._dev_exchange.nacc.item()) if self._dev_exchange is not None else 0)

    def _snapshot(self):
        self.engine.synchronize()
        t = self.t
        row = dict(n=t["n"].cpu().numpy(), vs=t["vs"].cpu().numpy().astype(np.float32),
                   z=t["z"].cpu().numpy().astype(np.float32), like=t["like"].cpu().numpy().astype(np.float32),
                   misfits=t["misfits"].cpu().numpy().astype(np.float32), noise=t["noise"].cpu().numpy().astype(np.float32),
                   vpvs=t["vpvs"].cpu().numpy().astype(np.float32),
                   beta=None if t["beta"] is None else t["beta"].cpu().numpy())
        self.snap["p1" if self.iiter < 0 else "p2"].append(row)
        # transdimensional chains: size the dispersion kernel's lane groups for the models the chains hold now
        self._hint = int(np.ceil(row["n"].mean()))

    def run(self, progress=None):
        self.snap_in_run = True
        try:
            while self.iiter < self.iter_phase2:
                if self.iiter % self.thinning == 0:
                    self._snapshot()
                before = self.iiter
                self.iterate()
                if progress is not None and self.iiter // 1000 != before // 1000:
                    progress(self)
        finally:
            self.snap_in_run = False
        self.engine.synchronize()
        return self

    # ---- results -------------------------------------------------------------------------------------
    def state_host(self):
        self.engine.synchronize()
        return {k: (None if v is None else v.cpu().numpy()) for k, v in self.t.items()}

    def samples(self, phase="p2", cold_only=False, gather=False):
        """Thinned samples: dict of arrays with leading axes [nsnap, C]; `models` in the reference's row layout
        [vs_1..vs_n NaN.., z_1..z_n NaN..] (2*maxlayers wide).
        gather: all chains of a sharded job (global chain order) instead of this rank's, on every rank.
        cold_only (tempered runs): one column per LADDER -- at every snapshot the state of the chain holding
        beta = 1; implies gather (the cold chain of a ladder moves between chains, hence between ranks);
        the ladder ids are returned as out["ladder"]."""
        S = self.snap[phase]
        ns, Cn, ML = len(S), self.C, self.ML
        models = np.full((ns, Cn, 2 * ML), np.nan, dtype=np.float32)
        j = np.arange(2 * ML)[:, None]                               # position inside a reference row
        for i, r in enumerate(S):
            n = r["n"][None, :].astype(np.int64)                      # [1, C]
            # reference rows hold the n vs values first, then the n depths, then NaN padding
            both = np.vstack((r["vs"], r["z"]))                       # [2*ML, C]: vs rows, then z rows
            src = np.where(j < n, j, ML + (j - n))                    # row of `both` feeding position j
            row = np.take_along_axis(both, np.clip(src, 0, 2 * ML - 1), axis=0)
            models[i] = np.where(j < 2 * n, row, np.nan).T
        out = dict(models=models)
        for k in ("like", "vpvs"):
            out[k + "s" if k == "like" else k] = np.array([r[k] for r in S], dtype=np.float32).reshape(ns, Cn)
        out["misfits"] = np.array([r["misfits"].T for r in S], dtype=np.float32).reshape(ns, Cn, self.nt + 1)
        out["noise"] = np.array([r["noise"].T for r in S], dtype=np.float32).reshape(ns, Cn, 2 * self.nt)
        if ns and S[0]["beta"] is not None:
            out["beta"] = np.array([r["beta"] for r in S]).reshape(ns, Cn)   # cold samples: out["beta"] == 1
        if not (gather or cold_only):
            return out
        from .parallel import gather_chain_axis, cold_samples
        dv = self.dev.index
        out = {k: gather_chain_axis(v, 1, self.dist, dv) for k, v in out.items()}
        out["chain_id"] = gather_chain_axis(self.chain_offset + np.arange(Cn, dtype=np.int64), 0, self.dist, dv)
        if len(np.unique(out["chain_id"])) != out["chain_id"].size:
            raise EngineError("sharded job with overlapping chain numbers (chain_offset): two ranks would draw the same "
                              "random streams and write the same files")
        if cold_only and "beta" in out:
            ladder = gather_chain_axis(self.ladder, 0, self.dist, dv)
            out.pop("chain_id")
            ids, out = cold_samples(out, ladder)
            out["ladder"] = ids
        return out

    def save(self, savepath=None):
        """c%03d_p{1,2}{models,likes,misfits,noise,vpvs}.npy, the reference's per-chain result files
        (src/SingleChain.py:646-690), written by rank 0 for ALL chains of the job with their global numbers
        (end-of-run all-gather of the thinned snapshots).  Tempered runs: one file set per ladder, holding the
        beta = 1 samples only (hot chains are not posterior samples)."""
        from .parallel import write_chain_files
        savepath = op.join(savepath or self.initparams["savepath"], "data")
        tempered = self.t["beta"] is not None
        for tag in ("p1", "p2"):
            if not self.snap[tag]:
                continue
            s = self.samples(tag, cold_only=tempered, gather=True)      # collective: every rank takes part
            if self.rank == 0:
                # file numbers = GLOBAL chain indices (= the chain's Philox / initial-state index), also with an
                # explicit chain_offset; tempered runs: ladder ids
                ids = s["ladder"] if tempered else s["chain_id"]
                write_chain_files(savepath, tag, s, ids)
        if self.rank == 0:
            from .results import save_config
            save_config(self.targets, op.join(savepath, "%s_config.pkl" % self.initparams.get("station", "test")),
                        priors=self.priors, initparams=self.initparams)
        if self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1:
            self.dist.barrier()
        return savepath
