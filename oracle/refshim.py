"""ctypes shims over oracle/_ref/*.so -- the UNMODIFIED reference sources compiled by
`make -C oracle ref` (only possible where /root/reference exists; the built .so files travel).

TEST INFRASTRUCTURE ONLY.  Used to (1) pin oracle/liboracle.so against the real reference,
(2) generate tests/golden/ fixtures, (3) optionally serve as bench.py's cpu_baseline with
kind="reference".  Never imported by bayhunter_amd/.

`surfdisp96` mimics the f2py call semantics (surf96_modsw.py:115-117: float64 arrays are
cast to float32 copies, `dispvel` float64 is written in place, `err` returned);
`synrf` mimics the Cython wrapper (rfmini.pyx:74-114 -> wrap.cpp:58-80).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")
_d = C.POINTER(C.c_double)
_f = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)

_surf = None
_rfm = None


def available():
    return os.path.exists(os.path.join(_REF, "libsurfdisp96.so")) and os.path.exists(
        os.path.join(_REF, "librfmini.so"))


def _surflib():
    global _surf
    if _surf is None:
        L = C.CDLL(os.path.join(_REF, "libsurfdisp96.so"))
        L.surfdisp96_.restype = None
        L.surfdisp96_.argtypes = [_f, _f, _f, _f, _ip, _ip, _ip, _ip, _ip, _ip, _d, _d, _ip]
        for name in ("dltar1_", "dltar4_"):
            fn = getattr(L, name)
            fn.restype = C.c_double
            # (wvno, omega, d, a, b, rho, rtp, dtp, btp, mmax, llw, twopi), all by reference
            fn.argtypes = [_d, _d, _f, _f, _f, _f, _f, _f, _f, _ip, _ip, _d]
        L.gtsolh_.restype = None
        L.gtsolh_.argtypes = [_f, _f, _f]
        _surf = L
    return _surf


def _rflib():
    global _rfm
    if _rfm is None:
        L = C.CDLL(os.path.join(_REF, "librfmini.so"))
        L.synrf_cwrap.restype = C.c_int
        L.synrf_cwrap.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_int, C.c_int, _d, _d, _d, _d, _d, _d,
                                  _d, _d, _d]
        _rfm = L
    return _rfm


def surfdisp96(thkm, vpm, vsm, rhom, nlayer, iflsph, iwave, mode, igr, kmax, t, cg):
    f = []
    for x in (thkm, vpm, vsm, rhom):
        buf = np.zeros(100, dtype=np.float32)
        x = np.asarray(x)
        buf[:x.size] = x.astype(np.float32)
        f.append(buf)
    tt = np.zeros(60)
    t = np.asarray(t, dtype=np.float64)
    tt[:t.size] = t
    out = np.zeros(60)
    ints = [C.c_int(int(v)) for v in (nlayer, iflsph, iwave, mode, igr, kmax)]
    err = C.c_int(0)
    _surflib().surfdisp96_(*[x.ctypes.data_as(_f) for x in f], *[C.byref(v) for v in ints],
                           tt.ctypes.data_as(_d), out.ctypes.data_as(_d), C.byref(err))
    cg[:min(cg.size, 60)] = out[:min(cg.size, 60)]
    return err.value


def dltar(wvno, omega, ifunc, d, a, b, rho):
    arrs = []
    for x in (d, a, b, rho):
        buf = np.zeros(100, dtype=np.float32)
        x = np.asarray(x, dtype=np.float32)
        buf[:x.size] = x
        arrs.append(buf)
    dummy = np.zeros(100, dtype=np.float32)
    mmax = C.c_int(len(d))
    llw = C.c_int(2 if arrs[2][0] <= 0.0 else 1)
    w, o, tp = C.c_double(wvno), C.c_double(omega), C.c_double(2 * np.pi)
    fn = _surflib().dltar1_ if ifunc == 1 else _surflib().dltar4_
    return fn(C.byref(w), C.byref(o), *[x.ctypes.data_as(_f) for x in arrs],
              dummy.ctypes.data_as(_f), dummy.ctypes.data_as(_f), dummy.ctypes.data_as(_f),
              C.byref(mmax), C.byref(llw), C.byref(tp))


def gtsolh(a, b):
    aa, bb, cc = C.c_float(a), C.c_float(b), C.c_float(0)
    _surflib().gtsolh_(C.byref(aa), C.byref(bb), C.byref(cc))
    return cc.value


def synrf(z, vp, vs, rh, qp, qs, p, a, nsamp, fsamp, tshift, nsv, sigma, wave):
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (z, vp, vs, rh, qp, qs)]
    nsamp = int(nsamp)
    waveno = {"P": 0, "SV": 1, "S": 1}[wave] if isinstance(wave, str) else int(wave)
    fz, fr, rf = np.zeros(nsamp), np.zeros(nsamp), np.zeros(nsamp)
    _rflib().synrf_cwrap(nsamp, float(fsamp), float(tshift), float(p), float(a), float(nsv),
                         float(sigma), waveno, arrs[0].size, *[x.ctypes.data_as(_d) for x in arrs],
                         fz.ctypes.data_as(_d), fr.ctypes.data_as(_d), rf.ctypes.data_as(_d))
    return fz, fr, rf
