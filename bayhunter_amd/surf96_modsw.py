"""Surface-wave dispersion plugin backed by the MI355X engine.

Host-side mirror of the reference plugin `SurfDisp` (src/surf96_modsw.py:13-126): same class
name, constructor, `set_modelparams`, `run_model(h, vp, vs, rho, **params) -> (x, y)` contract
and the same in-band failure convention (`(nan, nan)` when surf96 finds no root,
surf96_modsw.py:119-126).  The numerical work happens in bayhunter_amd/csrc/swd_kernel.hip
through the C ABI `bh_swd_batch`; `run_models` is the batched sibling the reference lacks.
"""
import numpy as np

from . import engine as _engine

# ref -> (iwave, igr); surf96_modsw.py:48-66
_SURFTAGS = {"rdispgr": (2, 1), "ldispgr": (1, 1), "rdispph": (2, 0), "ldispph": (1, 0)}
MAXPERIODS = _engine.MAX_PERIODS  # NP = 60, surfdisp96.f:61-62


class SurfDisp(object):
    def __init__(self, obsx, ref, engine=None):
        self.obsx = np.asarray(obsx, dtype=float)
        self.kmax = self.obsx.size
        self.ref = ref
        self.modelparams = {"mode": 1, "flsph": 0}
        self.wavetype, self.veltype = self.get_surftags(ref)
        self._engine = engine
        # more than 60 periods: compute on 60 evenly spaced periods and interpolate back
        # (surf96_modsw.py:35-43, :106-122)
        self.obsx_int = (np.linspace(self.obsx.min(), self.obsx.max(), MAXPERIODS)
                         if self.kmax > MAXPERIODS else None)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = _engine.default_engine()
        return self._engine

    def set_modelparams(self, **mparams):
        self.modelparams.update(mparams)

    @staticmethod
    def get_surftags(ref):
        try:
            return _SURFTAGS[ref]
        except KeyError:
            raise ReferenceError("Reference '%s' is not available in SurfDisp; available refs are "
                                 "rdispgr, ldispgr, rdispph, ldispph" % ref)

    def _periods(self):
        return self.obsx_int if self.obsx_int is not None else self.obsx

    def run_models(self, nlay, h, vp, vs, rho, layout="layer_major"):
        """Batch of models -> (x[K], y[B, K], err[B]); rows with err != 0 are NaN."""
        pers = self._periods()
        vel, err = self.engine.swd_batch(nlay, h, vp, vs, rho, pers, self.wavetype, self.veltype,
                                         mode=self.modelparams["mode"], flsph=self.modelparams["flsph"],
                                         layout=layout)
        if self.obsx_int is not None:
            vel = np.vstack([np.interp(self.obsx, pers, row) for row in vel])
        vel[err != 0] = np.nan
        return self.obsx, vel, err

    def run_model(self, h, vp, vs, rho, **params):
        h, vp, vs, rho = [np.asarray(a, dtype=float).reshape(-1, 1) for a in (h, vp, vs, rho)]
        nlay = np.array([h.shape[0]], dtype=np.int32)
        x, y, err = self.run_models(nlay, h, vp, vs, rho)
        if err[0] != 0:
            return np.nan, np.nan
        return x, y[0]
