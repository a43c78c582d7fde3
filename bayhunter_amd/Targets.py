"""Targets and the joint log-likelihood, evaluated on the MI355X engine.

Host-side mirror of the hot-path part of the reference's src/Targets.py: `ObservedData`,
`ModeledData`, `Valuation`, `SingleTarget` (+ the six concrete targets) and `JointTarget`
with `evaluate(h, vp, vs, noise, **kwargs)` -> `.proposallikelihood`, `.proposalmisfits`
(Targets.py:314-347), including the failure sentinel -1e15 / 1e15 (:325-328).  What differs:

* `JointTarget.evaluate` is ONE call into the C ABI (`bh_evaluate_batch`, batch of 1): forward
  models, RMS misfits and the closed-form likelihood all run on the GPU;
  `JointTarget.evaluate_batch` is the batched sibling used by the chain driver.
* The noise-covariance law of a target is whatever the sampler installed on it: the reference's
  `SingleChain.set_target_covariance` (src/SingleChain.py:159-205) assigns
  `target.get_covariance = target.valuation.get_covariance_{exp,nocorr,nocorr_scalederr,gauss}` and
  `JointTarget.evaluate` (Targets.py:335-337) calls it.  Here `SingleTarget.law()` reads that
  assignment back (which of the four accessors it is; for the Gauss law also the `corr_inv` /
  `logcorr_det` that `init_covariance_gauss` left in the valuation) and the engine evaluates the
  closed form of that law; a callable that is none of the four is an error, never a silent default.
  `select_noise_laws` / `set_noise_law` perform the same assignments.  The dense matrices the
  accessors return are kept for API compatibility and for tests; the engine never forms them.
* A user-supplied plugin (`target.update_plugin(obj)`, Targets.py:201-202, template
  templates/myfwd.py) still works: its synthetics are handed to `bh_loglike_batch`.
"""
import logging

import numpy as np

from . import engine as _engine
from .rfmini_modrf import RFminiModRF
from .surf96_modsw import SurfDisp

logger = logging.getLogger()

LAWS = {"nocorr": _engine.LAW_NOCORR, "nocorr_scalederr": _engine.LAW_NOCORR_SCALED,
        "exp": _engine.LAW_EXP, "gauss": _engine.LAW_GAUSS}
FAIL_LOGL, FAIL_MISFIT = -1e15, 1e15


class ObservedData(object):
    """x: monotone sample positions, y = y(x), optional yerr (Targets.py:16-30)."""

    def __init__(self, x, y, yerr=None):
        self.x = x
        self.y = y
        self.yerr = yerr
        if yerr is None or np.any(np.asarray(yerr) <= 0.) or np.any(np.isnan(yerr)):
            self.yerr = np.ones(np.size(x)) * np.nan


class ModeledData(object):
    """Holds the forward-modelling plugin of a target and its latest synthetics
    (Targets.py:33-82)."""

    def __init__(self, obsx, ref):
        if ref in ("prf", "srf"):
            self.plugin = RFminiModRF(obsx, ref)
            self.xlabel = "Time in s"
        elif ref in ("rdispph", "ldispph", "rdispgr", "ldispgr"):
            self.plugin = SurfDisp(obsx, ref)
            self.xlabel = "Period in s"
        else:
            logger.info("Please provide a forward modeling plugin for your target.\n"
                        "Use target.update_plugin(MyForwardClass())")
            self.plugin = None
            self.xlabel = "x"
        self.x = np.nan
        self.y = np.nan

    def update(self, plugin):
        self.plugin = plugin

    def calc_synth(self, h, vp, vs, **kwargs):
        rho = kwargs.pop("rho")
        self.x, self.y = self.plugin.run_model(h, vp, vs, rho=rho, **kwargs)


class Valuation(object):
    """RMS, covariance laws and log-likelihood in plain NumPy -- the dense formulation of the
    reference (Targets.py:85-183), kept for API compatibility.  Not used by the engine path."""

    def __init__(self):
        self.corr_inv = None
        self.logcorr_det = None
        self.misfit = None
        self.likelihood = None

    @staticmethod
    def get_rms(yobs, ymod):
        return np.sqrt(np.mean((ymod - yobs) ** 2))

    @staticmethod
    def get_covariance_nocorr(sigma, size, yerr=None, corr=0):
        return np.eye(size) / sigma ** 2, 2 * size * np.log(sigma)

    @staticmethod
    def get_covariance_nocorr_scalederr(sigma, size, yerr, corr=0):
        s = yerr / yerr.min()
        return np.diag(1.0 / (s * sigma ** 2)), 2 * size * np.log(sigma) + np.log(np.prod(s))

    @staticmethod
    def get_corr_inv(corr, size):
        diag = np.full(size, 1.0 + corr ** 2)
        diag[0] = diag[-1] = 1
        off = np.full(size - 1, -corr)
        return np.diag(diag) + np.diag(off, 1) + np.diag(off, -1)

    def get_covariance_exp(self, corr, sigma, size, yerr=None):
        c_inv = self.get_corr_inv(corr, size) / (sigma ** 2 * (1 - corr ** 2))
        return c_inv, 2 * size * np.log(sigma) + (size - 1) * np.log(1 - corr ** 2)

    def init_covariance_gauss(self, corr, size, rcond=None):
        """R_ij = corr^((i-j)^2); inverse and log-determinant once, on the host, with LAPACK,
        like Targets.py:150-160."""
        idx = np.arange(size)
        rmatrix = corr ** ((idx[:, None] - idx[None, :]).astype(float) ** 2)
        self.corr_inv = (np.linalg.pinv(rmatrix, rcond=rcond) if rcond is not None
                         else np.linalg.inv(rmatrix))
        self.logcorr_det = np.linalg.slogdet(rmatrix)[1]

    def get_covariance_gauss(self, sigma, size, yerr=None, corr=None):
        return self.corr_inv / sigma ** 2, 2 * size * np.log(sigma) + self.logcorr_det

    @staticmethod
    def get_likelihood(yobs, ymod, c_inv, logc_det):
        d = ymod - yobs
        return -0.5 * (yobs.size * np.log(2 * np.pi) + logc_det) - d.dot(c_inv).dot(d) / 2.


_ACCESSORS = {"get_covariance_nocorr": "nocorr", "get_covariance_nocorr_scalederr": "nocorr_scalederr",
              "get_covariance_exp": "exp", "get_covariance_gauss": "gauss"}


def _law_of_accessor(gc):
    """Which of Valuation's four covariance accessors `gc` is, or None.  A bound method of any
    Valuation object (this target's, a copy, an unpickled one) is compared through `__func__`, the two
    static accessors are the plain functions; a subclass override or any other callable is None."""
    fn = getattr(gc, "__func__", gc)
    for name, law in _ACCESSORS.items():
        ours = Valuation.__dict__[name]
        if fn is getattr(ours, "__func__", ours):
            return law
    return None


class SingleTarget(object):
    noiseref = "swd"

    def __init__(self, x, y, ref, yerr=None):
        self.ref = ref
        self.obsdata = ObservedData(x=x, y=y, yerr=yerr)
        self.moddata = ModeledData(obsx=x, ref=ref)
        self.valuation = Valuation()
        # The covariance accessor the sampler installs (SingleChain.py:159-205); `law()` derives the
        # engine's law from it.  `noise_law` remembers the last derived law, so that the reset to None
        # before pickling (utils.py:142-143) does not lose it; None = nothing installed yet.
        self.get_covariance = None
        self.noise_law = None
        logger.info("Initiated target: %s (ref: %s)" % (self.__class__.__name__, self.ref))

    def update_plugin(self, plugin):
        self.moddata.update(plugin)

    def set_noise_law(self, law, corr=None, rcond=None):
        """law in {'nocorr', 'nocorr_scalederr', 'exp', 'gauss'}; 'gauss' needs the fixed
        correlation `corr` (and optionally `rcond`) to pre-compute R^-1 and ln|R|."""
        if law not in LAWS:
            raise ValueError("unknown noise law %r" % law)
        self.noise_law = law
        v = self.valuation
        if law == "gauss":
            v.init_covariance_gauss(corr, self.obsdata.x.size, rcond=rcond)
        self.get_covariance = {"nocorr": v.get_covariance_nocorr,
                               "nocorr_scalederr": v.get_covariance_nocorr_scalederr,
                               "exp": v.get_covariance_exp, "gauss": v.get_covariance_gauss}[law]

    def law(self):
        """Name of the covariance law in force: the accessor installed in `get_covariance`
        (identified by identity with this valuation's four accessors, by name for an accessor of
        another Valuation instance, e.g. after unpickling), else the law remembered from the last
        installation.  Raises for a foreign callable and when no law was ever installed -- the
        reference fails in both cases too (Targets.py:335: it would call None / get other numbers)."""
        gc = self.get_covariance
        if gc is None:
            if self.noise_law is None:
                raise RuntimeError(
                    "target %r has no covariance law: assign target.get_covariance (as "
                    "SingleChain.set_target_covariance does) or call set_noise_law / select_noise_laws"
                    % self.ref)
            return self.noise_law
        law = _law_of_accessor(gc)
        if law is None:
            raise TypeError(
                "target %r: get_covariance = %r is none of Valuation.get_covariance_{nocorr,"
                "nocorr_scalederr,exp,gauss}; the engine evaluates these four laws in closed form and "
                "does not fall back to another one" % (self.ref, gc))
        if law == "gauss":
            v = gc.__self__
            if v.corr_inv is None or v.logcorr_det is None:
                raise RuntimeError("target %r: Gauss law without init_covariance_gauss" % self.ref)
            if v is not self.valuation:     # accessor bound to another valuation: take its R^-1
                self.valuation.corr_inv, self.valuation.logcorr_det = v.corr_inv, v.logcorr_det
            if np.shape(v.corr_inv) != (np.size(self.obsdata.x),) * 2:
                raise ValueError("target %r: corr_inv has shape %r for %d samples"
                                 % (self.ref, np.shape(v.corr_inv), np.size(self.obsdata.x)))
        self.noise_law = law
        return law

    def _moddata_valid(self):
        """Targets.py:204-214"""
        if not type(self.moddata.x) == np.ndarray:
            return False
        if not len(self.obsdata.x) == len(self.moddata.x):
            return False
        if not np.sum(self.obsdata.x - self.moddata.x) <= 1e-5:
            return False
        if not len(self.obsdata.y) == len(self.moddata.y):
            return False
        return True

    def calc_misfit(self):
        if not self._moddata_valid():
            self.valuation.misfit = FAIL_MISFIT
            return
        self.valuation.misfit = self.valuation.get_rms(self.obsdata.y, self.moddata.y)

    def calc_likelihood(self, c_inv, logc_det):
        if not self._moddata_valid():
            self.valuation.likelihood = FAIL_LOGL
            return
        self.valuation.likelihood = self.valuation.get_likelihood(
            self.obsdata.y, self.moddata.y, c_inv, logc_det)

    # -- engine side ---------------------------------------------------------------------------
    def engine_backed(self):
        return isinstance(self.moddata.plugin, (SurfDisp, RFminiModRF))

    def engine_desc(self):
        """Field dict for `bh_target_desc` (include/bh_engine.h)."""
        n = int(np.size(self.obsdata.x))
        law = self.law()
        d = {"law": LAWS[law], "n": n, "yobs": np.asarray(self.obsdata.y, dtype=float)}
        if law == "nocorr_scalederr":
            d["yerr"] = np.asarray(self.obsdata.yerr, dtype=float)
        if law == "gauss":
            d["rinv"] = np.ascontiguousarray(self.valuation.corr_inv, dtype=float)
            d["logdet_r"] = float(self.valuation.logcorr_det)
        p = self.moddata.plugin
        if isinstance(p, SurfDisp):
            d.update(kind=_engine.TARGET_SWD, iwave=p.wavetype, igr=p.veltype,
                     mode=p.modelparams["mode"], flsph=p.modelparams["flsph"],
                     x=np.asarray(self.obsdata.x, dtype=float))
        elif isinstance(p, RFminiModRF):
            a = p._call_args()
            d.update(kind=_engine.TARGET_RF, waveno=a["waveno"], nsamp=a["nsamp"], p=a["p"],
                     gauss=a["gauss"], fsamp=a["fsamp"], tshift=a["tshift"], nsv=a["nsv"])
        else:  # user plugin: only the likelihood part of the descriptor is used
            d.update(kind=_engine.TARGET_USER)
        return d


class RayleighDispersionPhase(SingleTarget):
    noiseref = "swd"

    def __init__(self, x, y, yerr=None):
        SingleTarget.__init__(self, x, y, "rdispph", yerr=yerr)


class RayleighDispersionGroup(SingleTarget):
    noiseref = "swd"

    def __init__(self, x, y, yerr=None):
        SingleTarget.__init__(self, x, y, "rdispgr", yerr=yerr)


class LoveDispersionPhase(SingleTarget):
    noiseref = "swd"

    def __init__(self, x, y, yerr=None):
        SingleTarget.__init__(self, x, y, "ldispph", yerr=yerr)


class LoveDispersionGroup(SingleTarget):
    noiseref = "swd"

    def __init__(self, x, y, yerr=None):
        SingleTarget.__init__(self, x, y, "ldispgr", yerr=yerr)


class PReceiverFunction(SingleTarget):
    noiseref = "rf"

    def __init__(self, x, y, yerr=None):
        SingleTarget.__init__(self, x, y, "prf", yerr=yerr)


class SReceiverFunction(SingleTarget):
    noiseref = "rf"

    def __init__(self, x, y, yerr=None):
        SingleTarget.__init__(self, x, y, "srf", yerr=yerr)


def select_noise_laws(targets, corrfix, noise_corr, rcond=None):
    """Choose each target's covariance law the way SingleChain.set_target_covariance does
    (src/SingleChain.py:159-205): correlation prior is a range -> exponential; fixed 0 ->
    diagonal (scaled by yerr if given); fixed non-zero -> Gaussian for RF, exponential else."""
    for target, fixed, corr in zip(targets, corrfix, noise_corr):
        if not fixed:
            target.set_noise_law("exp")
        elif corr == 0 and np.any(np.isnan(target.obsdata.yerr)):
            target.set_noise_law("nocorr")
        elif corr == 0:
            target.set_noise_law("nocorr_scalederr")
        elif target.noiseref == "rf":
            target.set_noise_law("gauss", corr=corr, rcond=rcond)
        else:
            target.set_noise_law("exp")


class JointTarget(object):
    """Joint likelihood of several targets (Targets.py:300-347), on the engine."""

    def __init__(self, targets, engine=None):
        self.targets = targets
        self.ntargets = len(targets)
        self._engine = engine
        self._registered = None
        self.proposallikelihood = None
        self.proposalmisfits = None

    @property
    def engine(self):
        if self._engine is None:
            self._engine = _engine.default_engine()
        return self._engine

    def get_misfits(self):
        misfits = [t.valuation.misfit for t in self.targets]
        return np.concatenate((misfits, [np.sum(misfits)]))

    def _signature(self):
        sig = []
        for t in self.targets:
            p = t.moddata.plugin
            sig.append((id(t), id(p), t.law(), id(t.valuation.corr_inv), t.valuation.logcorr_det,
                        tuple(sorted((k, str(v)) for k, v in getattr(p, "modelparams", {}).items()))))
        return tuple(sig)

    def _register(self):
        sig = self._signature()
        if self._registered != sig or self.engine._owner is not self:
            self.engine.set_targets([t.engine_desc() for t in self.targets])
            self.engine._owner = self
            self._registered = sig
        self._offsets = np.concatenate(([0], np.cumsum([np.size(t.obsdata.x) for t in self.targets])))

    def evaluate_batch(self, nlay, h, vp, vs, noise, rho=None, layout="layer_major", want_ymod=False):
        """B models at once: returns (logL[B], misfits[B, nt+1], err[B][, ymod])."""
        self._register()
        if all(t.engine_backed() for t in self.targets):
            return self.engine.evaluate_batch(nlay, h, vp, vs, noise, rho=rho, layout=layout,
                                              want_ymod=want_ymod)
        # mixed case: user plugins produce their synthetics on the host, model by model
        h, vp, vs = [np.asarray(a, dtype=float) for a in (h, vp, vs)]
        if layout == "model_major":
            h, vp, vs = h.T, vp.T, vs.T
            rho = None if rho is None else np.asarray(rho, dtype=float).T
        rho = vp * 0.32 + 0.77 if rho is None else np.asarray(rho, dtype=float)
        B = h.shape[1]
        ymod = np.zeros((B, self._offsets[-1]))
        fail = np.zeros((self.ntargets, B), dtype=np.int32)
        for it, t in enumerate(self.targets):
            lo, hi = self._offsets[it], self._offsets[it + 1]
            p = t.moddata.plugin
            if isinstance(p, SurfDisp):
                _, y, err = p.run_models(nlay, h, vp, vs, rho)
                ymod[:, lo:hi] = np.nan_to_num(y)
                fail[it] = err
            elif isinstance(p, RFminiModRF):
                ymod[:, lo:hi] = p.run_models(nlay, h, vp, vs, rho)[1]
            else:
                for b in range(B):
                    n = int(nlay[b])
                    x, y = p.run_model(h[:n, b], vp[:n, b], vs[:n, b], rho=rho[:n, b])
                    t.moddata.x, t.moddata.y = x, y
                    if t._moddata_valid():
                        ymod[b, lo:hi] = y
                    else:
                        fail[it, b] = 1
        out = self.engine.loglike_batch(ymod, noise, fail)
        return out + (ymod,) if want_ymod else out

    def evaluate(self, h, vp, vs, noise, **kwargs):
        """Single model, reference signature (Targets.py:314-347)."""
        rho = kwargs.pop("rho", None)
        h, vp, vs = [np.asarray(a, dtype=float) for a in (h, vp, vs)]
        if rho is None:
            rho = vp * 0.32 + 0.77
        nlay = np.array([h.size], dtype=np.int32)
        noise = np.asarray(noise, dtype=float).reshape(1, -1)[:, :2 * self.ntargets]
        logL, misfits, err, ymod = self.evaluate_batch(
            nlay, h.reshape(-1, 1), vp.reshape(-1, 1), vs.reshape(-1, 1), noise,
            rho=np.asarray(rho, dtype=float).reshape(-1, 1), want_ymod=True)
        if err[0] != 0:
            for t in self.targets:
                t.moddata.x, t.moddata.y = np.nan, np.nan
            self.proposallikelihood = FAIL_LOGL
            self.proposalmisfits = [FAIL_MISFIT] * (self.ntargets + 1)
            return
        for it, t in enumerate(self.targets):
            t.moddata.x = np.asarray(t.obsdata.x)
            t.moddata.y = ymod[0, self._offsets[it]:self._offsets[it + 1]].copy()
            t.valuation.misfit = misfits[0, it]
        self.proposallikelihood = float(logL[0])
        self.proposalmisfits = misfits[0].copy()
