"""Result store of an inversion in the reference's on-disk format, so that the reference's
`PlotFromStorage(configfile)` opens a folder written by this package:

    <savepath>/data/<station>_config.pkl      save_config          (src/utils.py:127-153, src/mcmcOptimizer.py:52-55)
    <savepath>/data/c%03d_p{1,2}*.npy         ChainBatch.save / DeviceChains.save  (src/SingleChain.py:646-690)
    <savepath>/data/outliers.dat              get_outliers         (src/Plotting.py:113-151)
    <savepath>/data/c_{models,likes,misfits,noise,vpvs}.npy   save_final_distribution (src/Plotting.py:157-262)

Host I/O only -- nothing here touches the GPU.
"""
import glob
import os
import os.path as op
import pickle
import sys

import numpy as np

# the reference's class of every object inside a pickled target (module path in the reference package)
_REF_CLASS = {
    "RayleighDispersionPhase": "BayHunter.Targets", "RayleighDispersionGroup": "BayHunter.Targets",
    "LoveDispersionPhase": "BayHunter.Targets", "LoveDispersionGroup": "BayHunter.Targets",
    "PReceiverFunction": "BayHunter.Targets", "SReceiverFunction": "BayHunter.Targets",
    "SingleTarget": "BayHunter.Targets", "ObservedData": "BayHunter.Targets", "ModeledData": "BayHunter.Targets",
    "Valuation": "BayHunter.Targets", "SurfDisp": "BayHunter.surf96_modsw", "RFminiModRF": "BayHunter.rfmini_modrf",
}
_RF_KEYS = {"z": "%.2f", "vp": "%.4f", "vs": "%.4f", "rho": "%.4f", "qp": "%.1f", "qs": "%.1f", "n": "%d"}


def _shadow(obj, classes):
    """A copy of `obj` as an instance of the reference's class of the same name (state = the attributes the
    reference's object holds: this package's classes mirror them; engine handles and bound methods are dropped)."""
    name = type(obj).__name__
    if name not in _REF_CLASS:
        return obj
    state = {}
    for k, v in obj.__dict__.items():
        if k.startswith("_") or k in ("noise_law", "get_covariance"):
            continue
        state[k] = _shadow(v, classes) if type(v).__name__ in _REF_CLASS else v
    if name in ("SingleTarget",) or _REF_CLASS[name] == "BayHunter.Targets" and "obsdata" in state:
        state["get_covariance"] = None       # the reference resets the bound method before pickling (utils.py:142-143)
    if name == "RFminiModRF":
        state.setdefault("keys", dict(_RF_KEYS))
    inst = object.__new__(classes[name])
    inst.__dict__.update(state)
    return inst


def _reference_classes():
    """name -> class object standing for the reference's class of that name while pickling.  Where the reference
    package itself is already imported its own classes are used; otherwise empty stand-in classes that carry the
    reference's (module, name) -- `_RefPickler` writes that name into the file (pickle stores classes BY NAME: the
    file then loads into the real classes wherever the reference is installed).  Nothing is registered in
    sys.modules: a concurrent `import BayHunter` in another thread never sees a stand-in."""
    classes = {}
    for name, mod in _REF_CLASS.items():
        m = sys.modules.get(mod)
        if m is not None and hasattr(m, name):
            classes[name] = getattr(m, name)
        else:
            classes[name] = type(name, (object,), {"__module__": mod, "_bh_ref_path": (mod, name)})
    return classes


class _RefPickler(pickle._Pickler):
    """The pure-Python pickler with one change: a stand-in class is written as the GLOBAL it stands for, without the
    importability check of `save_global` (the class lives in the reference package, which need not be installed
    where the file is written).  The dispatch table is this class's own copy with its own handler for classes:
    libraries that extend the stock table process-wide (dill does, once imported) do not change what is written."""
    dispatch = dict(pickle._Pickler.dispatch)

    def _save_class(self, obj):
        ref = obj.__dict__.get("_bh_ref_path")
        if ref is None:
            return pickle._Pickler.save_global(self, obj)
        self.write(pickle.GLOBAL + ref[0].encode("ascii") + b"\n" + ref[1].encode("ascii") + b"\n")
        self.memoize(obj)

    dispatch[type] = _save_class


def save_config(targets, configfile, priors=None, initparams=None):
    """Write the configuration pickle `PlotFromStorage.__init__` needs (src/Plotting.py:52-58): the targets (as
    instances of the reference's classes), their refs, priors and initparams (src/utils.py:127-153)."""
    tl = targets.targets if hasattr(targets, "targets") else list(targets)
    classes = _reference_classes()
    data = {"targets": [_shadow(t, classes) for t in tl], "targetrefs": [t.ref for t in tl],
            "priors": dict(priors or {}), "initparams": dict(initparams or {})}
    os.makedirs(op.dirname(op.abspath(configfile)), exist_ok=True)
    with open(configfile, "wb") as f:
        _RefPickler(f, protocol=2).dump(data)
    return configfile


def _chainfiles(datapath, phase, ftype):
    return sorted(glob.glob(op.join(datapath, "c???_p%d%s.npy" % (phase, ftype))))


def _chainidx(path):
    return int(op.basename(path)[1:4])


def get_outliers(datapath, dev=0.05):
    """Outlier chains by the median likelihood of the main phase relative to the best chain
    (src/Plotting.py:113-151); writes outliers.dat when there are any."""
    likefiles = _chainfiles(datapath, 2, "likes")
    chainidxs = np.array([_chainidx(f) for f in likefiles], dtype=float)
    chainmedians = np.array([np.median(np.load(f)) for f in likefiles])
    maxlike = np.max(chainmedians)
    if maxlike > 0:
        scores = chainmedians / maxlike
    elif maxlike < 0:
        scores = maxlike / chainmedians
    else:
        scores = np.ones_like(chainmedians)
    sel = np.where((1 - scores) > dev)
    outliers, outscores = chainidxs[sel], 1 - scores[sel]
    if len(outliers) > 0:
        with open(op.join(datapath, "outliers.dat"), "w") as f:
            f.write("# Outlier chainindices with %.3f deviation condition\n" % dev)
            for i, o in enumerate(outliers):
                f.write("%d\t%.3f\n" % (o, outscores[i]))
    return outliers


def save_final_distribution(datapath, maxmodels=200000, dev=0.05, rstate=None):
    """Merge the main-phase files of all non-outlier chains into c_{models,likes,misfits,noise,vpvs}.npy, the same
    number of models from every chain (src/Plotting.py:157-262).  `rstate`: the generator the reference's
    Plotting module draws the per-chain subsets from -- a module-level RandomState(333); a fresh one per call here,
    i.e. the files equal those of the reference's FIRST call in a process."""
    rstate = np.random.RandomState(333) if rstate is None else rstate
    outlierfile = op.join(datapath, "outliers.dat")
    if op.exists(outlierfile):
        os.remove(outlierfile)
    outliers = get_outliers(datapath, dev=dev)
    files = {k: _chainfiles(datapath, 2, k) for k in ("models", "likes", "misfits", "noise", "vpvs")}
    nchains = int(len(files["likes"]) - outliers.size)
    maxmodels = int(maxmodels)
    mpc = int(maxmodels / nchains)
    out = {k: None for k in files}
    alllikes = np.ones(maxmodels) * np.nan
    start = 0
    for i, lf in enumerate(files["likes"]):
        if _chainidx(lf) in outliers:
            continue
        n = len(np.load(lf))
        index = np.arange(n).astype(int)
        if n > mpc:
            index = rstate.choice(index, mpc, replace=False)
            index.sort()
        end = start + index.size
        for k in files:
            data = np.load(files[k][i])[index]
            if k == "likes":
                alllikes[start:end] = data
                continue
            if out[k] is None:
                out[k] = np.ones((maxmodels,) + data.shape[1:]) * np.nan
            out[k][start:end] = data
        start = end
    keep = ~np.isnan(alllikes)
    out["likes"] = alllikes
    for k in ("models", "likes", "misfits", "noise", "vpvs"):
        np.save(op.join(datapath, "c_%s" % k), out[k][keep])
    return outliers
