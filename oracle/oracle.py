"""ctypes front-end of oracle/liboracle.so -- the CPU restatement of the hot path.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package (bayhunter_amd/) must never import this module.

The call signatures mirror the reference's native entry points:
  surfdisp96(...)  <-> f2py `BayHunter.surfdisp96_ext.surfdisp96` (surf96_modsw.py:115-117)
  synrf(...)       <-> Cython `BayHunter.rfmini.synrf`             (rfmini_modrf.py:134-137)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_d = C.POINTER(C.c_double)
_f = C.POINTER(C.c_float)
_i32 = C.POINTER(C.c_int32)
_i64 = C.POINTER(C.c_int64)

LAW_NOCORR, LAW_NOCORR_SCALED, LAW_EXP, LAW_GAUSS = 0, 1, 2, 3


def build(force=False):
    """(Re)build liboracle.so with the Makefile next to this file."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("swd_oracle.c", "rf_oracle.c", "like_oracle.c", "oracle.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.bho_surfdisp96.restype = C.c_int
        L.bho_surfdisp96.argtypes = [_f, _f, _f, _f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, _d, _d, _i64]
        L.bho_dltar1.restype = C.c_double
        L.bho_dltar1.argtypes = [C.c_double, C.c_double, _f, _f, _f, C.c_int, C.c_int]
        L.bho_dltar4.restype = C.c_double
        L.bho_dltar4.argtypes = [C.c_double, C.c_double, _f, _f, _f, _f, C.c_int, C.c_int]
        L.bho_swd_set_search.restype = None
        L.bho_swd_set_search.argtypes = [C.c_int]
        L.bho_swd_set_scan.restype = None
        L.bho_swd_set_scan.argtypes = [C.c_int]
        L.bho_swd_set_group_split.restype = None
        L.bho_swd_set_group_split.argtypes = [C.c_int]
        L.bho_swd_set_scan_tuning.restype = None
        L.bho_swd_set_scan_tuning.argtypes = [C.c_int, C.c_int, C.c_int]
        L.bho_secular_vec.restype = None
        L.bho_secular_vec.argtypes = [C.c_int, C.c_double, C.c_double, _f, _f, _f, _f, C.c_int, C.c_int, _d]
        L.bho_swd_guarded_count.restype = C.c_int64
        L.bho_swd_guarded_count.argtypes = [C.c_int]
        L.bho_gtsolh.restype = C.c_float
        L.bho_gtsolh.argtypes = [C.c_float, C.c_float]
        L.bho_swd_batch.restype = None
        L.bho_swd_batch.argtypes = [C.c_int, C.c_int, _i32, _d, _d, _d, _d, C.c_int, _d, C.c_int,
                                    C.c_int, C.c_int, C.c_int, _d, _i32, _i64, C.c_int]
        L.bho_synrf.restype = C.c_int
        L.bho_synrf.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_double, C.c_int, C.c_int, _d, _d, _d, _d, _d, _d, _d]
        L.bho_rf_batch.restype = None
        L.bho_rf_batch.argtypes = [C.c_int, C.c_int, _i32, _d, _d, _d, _d, C.c_double, C.c_double,
                                   C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, _d, C.c_int]
        L.bho_loglike_dense.restype = C.c_double
        L.bho_loglike_dense.argtypes = [C.c_int, C.c_int, _d, _d, _d, C.c_double, C.c_double, _d,
                                        C.c_double]
        L.bho_rms.restype = C.c_double
        L.bho_rms.argtypes = [C.c_int, _d, _d]
        _LIB = L
    return _LIB


def _pd(a):
    return a.ctypes.data_as(_d)


def _pf(a):
    return a.ctypes.data_as(_f)


def surfdisp96(thkm, vpm, vsm, rhom, nlayer, iflsph, iwave, mode, igr, kmax, t, cg,
               return_neval=False):
    """Same calling convention as the f2py extension: model arrays are cast to float32
    copies, `t`/`cg` are float64, `cg` is written in place, `err` is returned."""
    f = [np.ascontiguousarray(np.asarray(x), dtype=np.float32) for x in (thkm, vpm, vsm, rhom)]
    t = np.ascontiguousarray(t, dtype=np.float64)
    assert cg.dtype == np.float64 and cg.flags.c_contiguous
    ne = C.c_int64(0)
    err = lib().bho_surfdisp96(_pf(f[0]), _pf(f[1]), _pf(f[2]), _pf(f[3]), int(nlayer),
                               int(iflsph), int(iwave), int(mode), int(igr), int(kmax), _pd(t),
                               _pd(cg), C.byref(ne))
    return (err, ne.value) if return_neval else err


def set_swd_search(fast):
    """1 / True: the engine's short root refinement as it is (a restatement of swd_common.h, NOT of the reference) instead of
    the reference's nevill; 2: the same with the guard the engine applies (models in the situations where the reference's
    own outcome hinges on the last bits of a root are re-run with the reference's sequence); 0: the reference's sequence.
    Process-wide.  Use `with swd_search(2): ...` in tests."""
    lib().bho_swd_set_search(int(fast))


def swd_guarded_count(reset=True):
    """models the guard of search mode 2 sent back to the reference sequence since the last reset"""
    return int(lib().bho_swd_guarded_count(1 if reset else 0))


class swd_scan:
    """with swd_scan(1): the engine's counted scan (same bits, fewer evaluations) instead of getsol's step-by-step scan"""

    def __init__(self, counted):
        self.counted = counted

    def __enter__(self):
        lib().bho_swd_set_scan(int(self.counted))

    def __exit__(self, *a):
        lib().bho_swd_set_scan(0)


class swd_group_split:
    """with swd_group_split(): fundamental-mode group velocities in the order the device runs them -- the chain of
    the first roots (t/(1+h)) over all periods, then the second roots (t/(1-h)) one by one, last period first -- instead of the
    reference's first / second / first / second ...: the same bits (tests/test_oracle_swd.py)"""

    def __enter__(self):
        lib().bho_swd_set_group_split(1)

    def __exit__(self, *a):
        lib().bho_swd_set_group_split(0)


def secular_vec(ifunc, omega, c, d, a, b, rho):
    """the surface vector of the reference-exact recursion (2 entries for Love, 5 for Rayleigh)"""
    f = [np.ascontiguousarray(x, dtype=np.float32) for x in (d, a, b, rho)]
    out = np.zeros(5)
    lib().bho_secular_vec(int(ifunc), float(omega), float(c), _pf(f[0]), _pf(f[1]), _pf(f[2]), _pf(f[3]), int(f[0].size), 1, _pd(out))
    return out[:2 if ifunc == 1 else 5]


class swd_search:
    def __init__(self, fast):
        self.fast = fast

    def __enter__(self):
        set_swd_search(self.fast)

    def __exit__(self, *a):
        set_swd_search(False)


def dltar(wvno, omega, ifunc, d, a, b, rho):
    d, a, b, rho = [np.ascontiguousarray(x, dtype=np.float32) for x in (d, a, b, rho)]
    mmax = d.size
    llw = 2 if b[0] <= 0.0 else 1
    if ifunc == 1:
        return lib().bho_dltar1(wvno, omega, _pf(d), _pf(b), _pf(rho), mmax, llw)
    return lib().bho_dltar4(wvno, omega, _pf(d), _pf(a), _pf(b), _pf(rho), mmax, llw)


def gtsolh(a, b):
    return float(lib().bho_gtsolh(np.float32(a), np.float32(b)))


def swd_batch(nlay, h, vp, vs, rho, periods, iwave, igr, mode=1, flsph=0, nthreads=0):
    """h, vp, vs, rho: [B, Lmax] float64.  Returns vel[B, K], err[B], total secular evals."""
    h, vp, vs, rho = [np.ascontiguousarray(x, dtype=np.float64) for x in (h, vp, vs, rho)]
    B, Lmax = h.shape
    nlay = np.ascontiguousarray(nlay, dtype=np.int32)
    periods = np.ascontiguousarray(periods, dtype=np.float64)
    K = periods.size
    vel = np.zeros((B, K))
    err = np.zeros(B, dtype=np.int32)
    ne = C.c_int64(0)
    lib().bho_swd_batch(B, Lmax, nlay.ctypes.data_as(_i32), _pd(h), _pd(vp), _pd(vs), _pd(rho), K,
                        _pd(periods), iwave, igr, mode, flsph, _pd(vel), err.ctypes.data_as(_i32),
                        C.byref(ne), nthreads)
    return vel, err, ne.value


def synrf(z, vp, vs, rh, qp, qs, p, a, nsamp, fsamp, tshift, nsv, sigma, wave):
    """Same argument order as the Cython `rfmini.synrf`; returns (None, None, rf)."""
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (z, vp, vs, rh, qp, qs)]
    nsamp = int(nsamp)
    waveno = {"P": 0, "SV": 1, "S": 1}[wave] if isinstance(wave, str) else int(wave)
    rf = np.zeros(nsamp)
    lib().bho_synrf(nsamp, float(fsamp), float(tshift), float(p), float(a), float(nsv),
                    float(sigma), waveno, arrs[0].size, *[_pd(x) for x in arrs], _pd(rf))
    return None, None, rf


def rf_batch(nlay, h, vp, vs, rho, p, gauss, nsamp, fsamp, tshift, waveno, nkeep, nthreads=0):
    h, vp, vs, rho = [np.ascontiguousarray(x, dtype=np.float64) for x in (h, vp, vs, rho)]
    B, Lmax = h.shape
    nlay = np.ascontiguousarray(nlay, dtype=np.int32)
    out = np.zeros((B, nkeep))
    lib().bho_rf_batch(B, Lmax, nlay.ctypes.data_as(_i32), _pd(h), _pd(vp), _pd(vs), _pd(rho),
                       float(p), float(gauss), int(nsamp), float(fsamp), float(tshift), int(waveno),
                       int(nkeep), _pd(out), nthreads)
    return out


def loglike_dense(law, ymod, yobs, corr, sigma, yerr=None, rinv=None, logdet_r=0.0):
    ymod = np.ascontiguousarray(ymod, dtype=np.float64)
    yobs = np.ascontiguousarray(yobs, dtype=np.float64)
    n = ymod.size
    yerr_p = _pd(np.ascontiguousarray(yerr, dtype=np.float64)) if yerr is not None else None
    rinv_c = np.ascontiguousarray(rinv, dtype=np.float64) if rinv is not None else None
    rinv_p = _pd(rinv_c) if rinv_c is not None else None
    return lib().bho_loglike_dense(law, n, _pd(ymod), _pd(yobs), yerr_p, float(corr), float(sigma),
                                   rinv_p, float(logdet_r))


class Target(C.Structure):
    _fields_ = [("kind", C.c_int32), ("law", C.c_int32), ("n", C.c_int32), ("iwave", C.c_int32),
                ("igr", C.c_int32), ("waveno", C.c_int32), ("nsamp", C.c_int32),
                ("p_s_per_deg", C.c_double), ("gauss", C.c_double), ("fsamp", C.c_double),
                ("tshift", C.c_double), ("x", _d), ("yobs", _d), ("yerr", _d)]


def joint_batch(nlay, h, vp, vs, rho, targets, noise, nthreads=0):
    """targets: list of dicts (kind, law, n, iwave, igr, waveno, nsamp, p, gauss, fsamp, tshift,
    x, yobs, yerr).  h.. are [B, Lmax].  Returns (logL[B], misfits[B, nt+1])."""
    h, vp, vs, rho = [np.ascontiguousarray(a, dtype=np.float64) for a in (h, vp, vs, rho)]
    B, Lmax = h.shape
    nlay = np.ascontiguousarray(nlay, dtype=np.int32)
    nt = len(targets)
    arr = (Target * nt)()
    keep = []
    for i, d in enumerate(targets):
        t = arr[i]
        t.kind, t.law, t.n = int(d["kind"]), int(d["law"]), int(d["n"])
        t.iwave, t.igr = int(d.get("iwave", 2)), int(d.get("igr", 0))
        t.waveno, t.nsamp = int(d.get("waveno", 0)), int(d.get("nsamp", 0))
        t.p_s_per_deg, t.gauss = float(d.get("p", 6.4)), float(d.get("gauss", 1.0))
        t.fsamp, t.tshift = float(d.get("fsamp", 1.0)), float(d.get("tshift", 0.0))
        for key in ("x", "yobs", "yerr"):
            v = d.get(key)
            if v is not None:
                v = np.ascontiguousarray(v, dtype=np.float64)
                keep.append(v)
                setattr(t, key, _pd(v))
    noise = np.ascontiguousarray(noise, dtype=np.float64)
    logL = np.zeros(B)
    misf = np.zeros((B, nt + 1))
    fn = lib().bho_joint_batch
    fn.restype = None
    fn.argtypes = [C.c_int, C.c_int, _i32, _d, _d, _d, _d, C.c_int, C.POINTER(Target), _d, _d, _d, C.c_int]
    fn(B, Lmax, nlay.ctypes.data_as(_i32), _pd(h), _pd(vp), _pd(vs), _pd(rho), nt, arr, _pd(noise),
       _pd(logL), _pd(misf), int(nthreads))
    return logL, misf


def libm_probe(op, x):
    """op 0 sin, 1 cos (via sincos(), like the reference's compiled code), 2 exp -- the host libm."""
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    out = np.zeros_like(x)
    fn = lib().bho_libm_probe
    fn.restype = None
    fn.argtypes = [C.c_int, C.c_int, _d, _d]
    fn(int(op), x.size, _pd(x), _pd(out))
    return out


def rms(ymod, yobs):
    ymod = np.ascontiguousarray(ymod, dtype=np.float64)
    yobs = np.ascontiguousarray(yobs, dtype=np.float64)
    return lib().bho_rms(ymod.size, _pd(ymod), _pd(yobs))
