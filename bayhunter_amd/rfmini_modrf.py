"""Receiver-function plugin backed by the MI355X engine.

Host-side mirror of the reference plugin `RFminiModRF` (src/rfmini_modrf.py:13-154): same
constructor, `set_modelparams`, `run_model(h, vp, vs, rho, **params) -> (time, rf)`.  The
sampling parameters are derived from the observed time axis exactly as
rfmini_modrf.py:41-62 does.  The synthesis runs in bayhunter_amd/csrc/rf_kernel.hip through
the C ABI `bh_rf_batch`; `run_models` is the batched sibling.
"""
import numpy as np

from . import engine as _engine


class RFminiModRF(object):
    def __init__(self, obsx, ref, engine=None):
        self.ref = ref
        self.obsx = np.asarray(obsx, dtype=float)
        self._init_obsparams()
        if ref in ("prf", "seis"):
            self.modelparams = {"wtype": "P"}
        elif ref == "srf":
            self.modelparams = {"wtype": "SV"}
        else:
            raise ReferenceError("Reference '%s' is not available in RFminiModRF (prf, srf)" % ref)
        # `water` is accepted for compatibility; the reference's C++ never applies it
        # (rfmini/greens.cpp:384 is commented out)
        self.modelparams.update({"gauss": 1.0, "p": 6.4, "water": 0.001, "nsv": None})
        self._engine = engine

    @property
    def engine(self):
        if self._engine is None:
            self._engine = _engine.default_engine()
        return self._engine

    MAX_NSAMP = 1 << 18   # longest transform of the synthesis kernel (csrc/bh_device.h: BH_RF_MAX_NSAMP)

    def _init_obsparams(self):
        """fsamp [Hz], tshft [s], nsamp (power of two >= 2*ndata); rfmini_modrf.py:41-62."""
        steps = np.round(np.diff(self.obsx), 4)
        if np.unique(steps).size != 1:
            raise ValueError("Target: %s. Sampling rate must be constant." % self.ref)
        self.fsamp = 1.0 / float(steps[0])
        self.tshft = -self.obsx[0]
        self.nsamp = 2 ** int(np.ceil(np.log2(self.obsx.size * 2)))
        # The reference has no bound here (rfmini_modrf.py:62; fork.cpp:11-60 transforms any power of two).  The engine's synthesis
        # kernel keeps the spectrum of a trace in ONE workgroup's LDS up to nsamp = 16384 (observed traces of up to 8192 samples) and in
        # an HBM workspace beyond, up to MAX_NSAMP = 262144 (131072 observed samples); longer ones are refused here, by name, before
        # any native call (the C ABI answers BH_EUNSUPPORTED).
        if self.nsamp > self.MAX_NSAMP:
            raise ValueError("Target: %s. %d observed samples need a transform of %d points; the MI355X engine's receiver-function "
                             "kernel holds at most MAX_NSAMP = %d (observed traces of up to %d samples)."
                             % (self.ref, self.obsx.size, self.nsamp, self.MAX_NSAMP, self.MAX_NSAMP // 2))

    def set_modelparams(self, **mparams):
        self.modelparams.update(mparams)

    def _call_args(self):
        mp = self.modelparams
        waveno = {"P": _engine.RF_P, "SV": _engine.RF_SV}[mp["wtype"]]
        nsv = 0.0 if mp["nsv"] is None else float(mp["nsv"])
        return dict(p=mp["p"], gauss=mp["gauss"], nsamp=self.nsamp, fsamp=self.fsamp,
                    tshift=self.tshft, waveno=waveno, nkeep=self.obsx.size, nsv=nsv)

    def time_axis(self):
        return (np.arange(self.nsamp) / self.fsamp - self.tshft)[:self.obsx.size]

    def run_models(self, nlay, h, vp, vs, rho, qp=None, qs=None, layout="layer_major"):
        """Batch of models -> (time[n], rf[B, n])."""
        rf = self.engine.rf_batch(nlay, h, vp, vs, rho, qp=qp, qs=qs, layout=layout, **self._call_args())
        return self.time_axis(), rf

    def compute_rf(self, h, vp, vs, rho, **params):
        cols = [np.asarray(a, dtype=float).reshape(-1, 1) for a in (h, vp, vs, rho)]
        qp, qs = params.get("qp"), params.get("qs")
        qp = None if qp is None else np.asarray(qp, dtype=float).reshape(-1, 1)
        qs = None if qs is None else np.asarray(qs, dtype=float).reshape(-1, 1)
        if (qp is None) != (qs is None):  # one given: the other takes the reference default
            n = cols[0].shape[0]
            qp = np.full((n, 1), 500.0) if qp is None else qp
            qs = np.full((n, 1), 225.0) if qs is None else qs
        nlay = np.array([cols[0].shape[0]], dtype=np.int32)
        t, rf = self.run_models(nlay, *cols, qp=qp, qs=qs)
        return t, rf[0]

    def run_model(self, h, vp, vs, rho, **params):
        h, vp, vs, rho = [np.asarray(a) for a in (h, vp, vs, rho)]
        assert h.size == vp.size == vs.size == rho.size
        return self.compute_rf(h, vp, vs, rho, **params)
