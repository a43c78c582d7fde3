"""Model parametrisation helpers: Voronoi nuclei -> layered (vp, vs, h).

Mirror of the part of the reference's `Model` that feeds the hot path
(src/Models.py:16-52: split_modelparams, get_vp, get_vp_vs_h) plus a batched packer that
turns many nuclei vectors into the layer-major arrays the engine takes.
"""
import numpy as np


class Model(object):
    @staticmethod
    def split_modelparams(model):
        """[vs_1..vs_n, z_1..z_n] (NaN padded) -> (n, vs, z_vnoi); Models.py:16-24."""
        model = np.asarray(model, dtype=float)
        model = model[~np.isnan(model)]
        n = model.size // 2
        return n, model[:n], model[-n:]

    @staticmethod
    def get_vp(vs, vpvs=1.73, mantle=(4.3, 1.8)):
        """Crustal vp/vs down to the first layer with vs >= mantle[0], mantle[1] below;
        Models.py:26-37."""
        vs = np.asarray(vs, dtype=float)
        vp = vs * vpvs
        deep = np.flatnonzero(vs >= mantle[0])
        if deep.size:
            vp[deep[0]:] = vs[deep[0]:] * mantle[1]
        return vp

    @staticmethod
    def get_vp_vs_h(model, vpvs=1.73, mantle=None):
        """Layer interfaces at the mid-points of consecutive nuclei; Models.py:39-52."""
        n, vs, z_vnoi = Model.split_modelparams(model)
        z_disc = (z_vnoi[:n - 1] + z_vnoi[1:n]) / 2.
        h = np.concatenate((z_disc - np.concatenate(([0], z_disc[:-1])), [0]))
        vp = Model.get_vp(vs, vpvs, mantle) if mantle is not None else vs * vpvs
        return vp, vs, h

    @staticmethod
    def pack_batch(models, vpvs, mantle=None, Lmax=None):
        """Many nuclei vectors -> (nlay[B], h, vp, vs) layer-major [Lmax, B] for the engine.
        `vpvs` may be a scalar or one value per model."""
        B = len(models)
        vpvs = np.broadcast_to(np.asarray(vpvs, dtype=float), (B,))
        rows = [Model.get_vp_vs_h(m, k, mantle) for m, k in zip(models, vpvs)]
        nlay = np.array([r[1].size for r in rows], dtype=np.int32)
        Lmax = int(nlay.max()) if Lmax is None else Lmax
        h = np.zeros((Lmax, B)); vp = np.zeros((Lmax, B)); vs = np.zeros((Lmax, B))
        for b, (rvp, rvs, rh) in enumerate(rows):
            n = nlay[b]
            vp[:n, b], vs[:n, b], h[:n, b] = rvp, rvs, rh
        return nlay, h, vp, vs


class ModelMatrix(object):
    """The one helper of the reference's `ModelMatrix` the result store needs (src/Models.py:227-274; the
    depth-interpolation / histogram helpers of that class serve plotting only and stay with the reference)."""

    @staticmethod
    def get_weightedvalues(weights, models=None, likes=None, misfits=None, noiseparams=None, vpvs=None):
        """Rows repeated by `weights` (the number of iterations a model stayed current): returns
        (wmodels, wlikes, wmisfits, wnoise, wvpvs), None for what was not given -- same outputs as the
        reference's loops, as one `np.repeat` each."""
        weights = np.array(weights, dtype=int)

        def rep(a, as_rows):
            if a is None:
                return None
            a = np.asarray(a, dtype=float)
            if as_rows and a.ndim == 1:      # the reference treats scalars per model as a vector ...
                return np.repeat(a, weights)
            return np.repeat(a, weights, axis=0)

        return (rep(models, False), rep(likes, True), rep(misfits, True), rep(noiseparams, False), rep(vpvs, True))
