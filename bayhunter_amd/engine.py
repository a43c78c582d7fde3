"""ctypes binding of the engine's C ABI (include/bh_engine.h -> bayhunter_amd/libbh_engine.so).

This is the only place Python touches the native library.  There is NO CPU fallback: if the
shared library is missing, or no MI355X/HIP device is usable, creating an `Engine` raises
`EngineError`.  (The CPU restatement under oracle/ is test infrastructure and is never
imported from here.)

Host API  : numpy arrays in, numpy arrays out (the library stages through its own device buffers).
Device API: raw device pointers (`tensor.data_ptr()`), asynchronous on a HIP stream -- what
            bench.py and the multi-GPU chain driver use with torch-owned HBM buffers.
"""
import contextlib
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbh_engine.so")

BH_OK, BH_EINVAL, BH_EHIP, BH_ENOMEM, BH_EUNSUPPORTED = 0, -1, -2, -3, -4
HOST, DEVICE = 0, 1
WAVE_LOVE, WAVE_RAYLEIGH = 1, 2
VEL_PHASE, VEL_GROUP = 0, 1
RF_P, RF_SV = 0, 1
LAW_NOCORR, LAW_NOCORR_SCALED, LAW_EXP, LAW_GAUSS = 0, 1, 2, 3
TARGET_SWD, TARGET_RF, TARGET_USER = 0, 1, 2
SEARCH_REFERENCE, SEARCH_FAST, SEARCH_FAST_RAYLEIGH = 0, 1, 2  # bh_engine_set_swd_search
ARITH_EXACT, ARITH_FAST = 0, 1  # bh_engine_set_swd_arith
SCAN_STEPS, SCAN_COUNTED, SCAN_AUTO = 0, 1, 2  # bh_engine_set_swd_scan
MAX_PERIODS, MAX_LAYERS, MAX_TARGETS = 60, 100, 8

_d = C.POINTER(C.c_double)
_i32 = C.POINTER(C.c_int32)


class EngineError(RuntimeError):
    pass


class TargetDesc(C.Structure):
    """Mirror of `bh_target_desc` (include/bh_engine.h)."""
    _fields_ = [("kind", C.c_int32), ("law", C.c_int32), ("n", C.c_int32),
                ("iwave", C.c_int32), ("igr", C.c_int32), ("mode", C.c_int32), ("flsph", C.c_int32),
                ("waveno", C.c_int32), ("nsamp", C.c_int32),
                ("p_s_per_deg", C.c_double), ("gauss", C.c_double), ("fsamp", C.c_double),
                ("tshift", C.c_double), ("nsv", C.c_double),
                ("x", _d), ("yobs", _d), ("yerr", _d), ("rinv", _d), ("logdet_r", C.c_double)]


BH_MAX_TARGETS = 8
BH_CHAIN_MAXLAYERS = 32
BH_CHAIN_MAXDEPTH = 7


class ChainConfig(C.Structure):
    """bh_chain_config of include/bh_engine.h"""
    _fields_ = [("nt", C.c_int32), ("maxlayers", C.c_int32), ("layermin", C.c_int32), ("layermax", C.c_int32),
                ("iter_burnin", C.c_int32), ("iterations", C.c_int32),
                ("vsmin", C.c_double), ("vsmax", C.c_double), ("zmin", C.c_double), ("zmax", C.c_double),
                ("thickmin", C.c_double), ("lvz", C.c_double), ("hvz", C.c_double),
                ("vpvsmin", C.c_double), ("vpvsmax", C.c_double), ("mantle_vs", C.c_double), ("mantle_vpvs", C.c_double),
                ("acc_lo", C.c_double), ("acc_hi", C.c_double),
                ("noise_lo", C.c_double * (2 * BH_MAX_TARGETS)), ("noise_hi", C.c_double * (2 * BH_MAX_TARGETS)),
                ("seed", C.c_uint64), ("chain_offset", C.c_int64)]


CHAIN_STATE_FIELDS = ("n", "vs", "z", "vpvs", "noise", "like", "misfits", "propdist", "proposed", "accepted", "naccepted",
                      "beta", "pn", "move", "valid", "pvs", "pz", "pvpvs", "pnoise", "dvs2", "lay_n", "lay_h", "lay_vp", "lay_vs",
                      "inject", "lay_rho")


class ChainState(C.Structure):
    """bh_chain_state of include/bh_engine.h (all members are device pointers)"""
    _fields_ = [(k, C.c_void_p) for k in CHAIN_STATE_FIELDS]


_lib = None


def load_library():
    """dlopen libbh_engine.so once.  torch (when the process uses it) must be imported BEFORE
    the library so that both share one HIP runtime (torch ships its own libamdhip64.so with the
    same SONAME); importing it here first makes the order irrelevant for callers."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError("native engine %s is not built; run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (or `make -C bayhunter_amd/csrc`)" % LIB_PATH)
    if "torch" not in sys.modules and os.environ.get("BH_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional for the plugin path
            pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    L.bh_abi_version.restype = C.c_int
    L.bh_engine_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.bh_engine_destroy.argtypes = [vp]
    L.bh_engine_destroy.restype = None
    L.bh_engine_last_error.argtypes = [vp]
    L.bh_engine_last_error.restype = C.c_char_p
    L.bh_engine_stream.argtypes = [vp]
    L.bh_engine_stream.restype = vp
    L.bh_engine_synchronize.argtypes = [vp]
    L.bh_engine_set_instrumentation.argtypes = [vp, C.c_int, C.c_int]
    L.bh_engine_set_swd_group.argtypes = [vp, C.c_int]
    L.bh_engine_set_swd_lookahead.argtypes = [vp, C.c_int]
    L.bh_engine_set_swd_search.argtypes = [vp, C.c_int]
    L.bh_engine_get_swd_search.argtypes = [vp]
    L.bh_engine_set_swd_arith.argtypes = [vp, C.c_int]
    L.bh_engine_last_swd_kernel.argtypes = [vp]
    L.bh_engine_set_swd_trials.argtypes = [vp, C.c_int]
    L.bh_engine_get_swd_trials.argtypes = [vp]
    L.bh_engine_get_swd_arith.argtypes = [vp]
    L.bh_engine_set_swd_scan.argtypes = [vp, C.c_int]
    L.bh_engine_get_swd_scan.argtypes = [vp]
    L.bh_engine_set_tuning.argtypes = [vp, C.c_char_p, C.c_int]
    L.bh_engine_get_tuning.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    L.bh_engine_guard_stats.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.bh_engine_set_typical_layers.argtypes = [vp, C.c_int]
    L.bh_engine_set_model_order.argtypes = [vp, C.c_int]
    L.bh_timing_reset.argtypes = [vp]
    L.bh_timing_collect.argtypes = [vp, C.POINTER(C.c_int), _d, _d]
    L.bh_timing_steps.argtypes = [vp, C.c_int, _d, C.POINTER(C.c_int)]
    L.bh_last_neval.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.bh_debug_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.bh_debug_trace.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.bh_swd_batch.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_ssize_t,
                               C.c_ssize_t, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.bh_rf_batch.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                              C.c_ssize_t, C.c_ssize_t, C.c_double, C.c_double, C.c_int, C.c_double,
                              C.c_double, C.c_double, C.c_int, C.c_int, vp]
    L.bh_targets_set.argtypes = [vp, C.c_int, C.POINTER(TargetDesc)]
    L.bh_evaluate_batch.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp,
                                    C.c_ssize_t, C.c_ssize_t, vp, vp, vp, vp, vp]
    L.bh_loglike_batch.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.bh_probe_math.argtypes = [vp, C.c_int, C.c_int, _d, _d]
    L.bh_chain_propose.argtypes = [vp, C.POINTER(ChainConfig), C.POINTER(ChainState), C.c_int, C.c_int]
    L.bh_chain_accept.argtypes = [vp, C.POINTER(ChainConfig), C.POINTER(ChainState), C.c_int, C.c_int, vp, vp]
    L.bh_chain_propose_window.argtypes = [vp, C.POINTER(ChainConfig), C.POINTER(ChainState), C.c_int, C.c_int, C.c_int, C.c_ssize_t]
    L.bh_chain_accept_window.argtypes = [vp, C.POINTER(ChainConfig), C.POINTER(ChainState), C.c_int, C.c_int, C.c_int, C.c_ssize_t, vp, vp]
    for name in ("bh_engine_create", "bh_engine_synchronize", "bh_engine_set_instrumentation", "bh_engine_set_swd_group", "bh_engine_set_swd_lookahead", "bh_engine_set_swd_search", "bh_engine_get_swd_search", "bh_engine_set_swd_arith", "bh_engine_get_swd_arith", "bh_engine_last_swd_kernel", "bh_engine_set_swd_trials", "bh_engine_get_swd_trials", "bh_engine_set_swd_scan", "bh_engine_get_swd_scan", "bh_engine_set_tuning", "bh_engine_get_tuning", "bh_engine_guard_stats", "bh_engine_set_typical_layers", "bh_engine_set_model_order",
                 "bh_timing_reset", "bh_timing_collect", "bh_timing_steps", "bh_last_neval", "bh_debug_counters", "bh_debug_trace", "bh_swd_batch", "bh_rf_batch", "bh_targets_set",
                 "bh_evaluate_batch", "bh_loglike_batch", "bh_probe_math", "bh_chain_propose", "bh_chain_accept",
                 "bh_chain_propose_window", "bh_chain_accept_window"):
        getattr(L, name).restype = C.c_int
    if L.bh_abi_version() != 10:
        raise EngineError("ABI version mismatch")
    _lib = L
    return L


# include/bh_engine.h: the drop-in contract
EXPORTED_SYMBOLS = ("bh_abi_version", "bh_engine_create", "bh_engine_destroy", "bh_engine_last_error",
                    "bh_engine_stream", "bh_engine_synchronize", "bh_engine_set_swd_search", "bh_engine_get_swd_search",
                    "bh_engine_set_swd_arith", "bh_engine_get_swd_arith", "bh_engine_set_swd_trials", "bh_engine_get_swd_trials",
                    "bh_engine_guard_stats", "bh_engine_set_typical_layers", "bh_engine_set_model_order",
                    "bh_swd_batch", "bh_rf_batch", "bh_targets_set", "bh_evaluate_batch", "bh_loglike_batch",
                    "bh_chain_propose", "bh_chain_accept", "bh_chain_propose_window", "bh_chain_accept_window")
# include/bh_engine_debug.h: measurement, diagnostics, experiment switches (bench.py, tools/, tests)
DEBUG_SYMBOLS = ("bh_engine_set_swd_group", "bh_engine_set_swd_lookahead", "bh_engine_last_swd_kernel", "bh_engine_set_swd_scan",
                 "bh_engine_get_swd_scan", "bh_engine_set_tuning",
                 "bh_engine_get_tuning", "bh_probe_math", "bh_engine_set_instrumentation", "bh_timing_reset",
                 "bh_timing_collect", "bh_timing_steps", "bh_last_neval", "bh_debug_counters", "bh_debug_trace")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def pack_models(models, Lmax=None):
    """List of (h, vp, vs, rho) 1-D arrays -> layer-major float64 [Lmax, B] arrays + nlay[B]."""
    B = len(models)
    nlay = np.array([len(m[0]) for m in models], dtype=np.int32)
    if Lmax is None:
        Lmax = int(nlay.max()) if B else 1
    out = [np.zeros((Lmax, B)) for _ in range(4)]
    for b, m in enumerate(models):
        for a, v in zip(out, m):
            a[:nlay[b], b] = v
    return nlay, out[0], out[1], out[2], out[3]


class Engine(object):
    """One engine = one GPU (HIP device ordinal `device`) = one stream."""

    def __init__(self, device=0):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.bh_engine_create(int(device), C.byref(h))
        if rc != BH_OK:
            raise EngineError("bh_engine_create(device=%d) failed with %d: no usable HIP device "
                              "(this package has no CPU fallback)" % (device, rc))
        self._h = h
        self.device = int(device)
        self._owner = None  # the JointTarget whose targets are currently registered
        self.ntargets = 0
        self.ldy = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.bh_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != BH_OK:
            msg = self._L.bh_engine_last_error(self._h)
            raise EngineError("engine call failed (%d): %s" % (rc, msg.decode() if msg else "?"))

    # -- plumbing ---------------------------------------------------------------------------
    @property
    def stream(self):
        return self._L.bh_engine_stream(self._h)

    def synchronize(self):
        self._check(self._L.bh_engine_synchronize(self._h))

    def set_instrumentation(self, timing=False, counting=False):
        self._check(self._L.bh_engine_set_instrumentation(self._h, int(timing), int(counting)))

    def set_swd_group(self, lanes_per_model):
        """0 = automatic; 1..32 lanes of a wavefront per model in the dispersion kernel."""
        self._check(self._L.bh_engine_set_swd_group(self._h, int(lanes_per_model)))

    def set_typical_layers(self, nlay):
        """Hint for device-resident batches: typical layer count (0 = unknown)."""
        self._check(self._L.bh_engine_set_typical_layers(self._h, int(nlay)))

    def set_model_order(self, sort_by_depth=True):
        """False: batches are of uniform depth (or sorted already): skip the on-device sort by layer count."""
        self._check(self._L.bh_engine_set_model_order(self._h, int(bool(sort_by_depth))))

    def set_swd_lookahead(self, trials_per_round):
        """0 = automatic; 1..12 trial phase velocities per round of the dispersion root search."""
        self._check(self._L.bh_engine_set_swd_lookahead(self._h, int(trials_per_round)))

    def set_swd_search(self, search):
        """"fast" (the engine's default): the reference's brackets, ~3 evaluations inside each instead of nevill's 10-12;
        fundamental-mode phase-velocity targets only; within 1.2e-6 relative of the reference (north_star: 1e-5), failure
        flags the reference's (a guard re-runs the models whose outcome hinges on the last bits of a root with the
        reference's sequence).  "reference": the reference's sequence of secular-function evaluations, velocities
        bit-identical to surfdisp96 (replays of recorded reference chains: ChainBatch).  "fast_rayleigh": "fast" for the
        Rayleigh targets only -- for samplers, whose Love proposals trip the guard too often (include/bh_engine.h)."""
        codes = {"reference": SEARCH_REFERENCE, "fast": SEARCH_FAST, "fast_rayleigh": SEARCH_FAST_RAYLEIGH}
        codes.update({v: v for v in list(codes.values())})
        code = codes.get(search)
        if code is None:
            raise ValueError("search must be 'reference', 'fast' or 'fast_rayleigh'")
        self._check(self._L.bh_engine_set_swd_search(self._h, code))

    def swd_search(self):
        return {SEARCH_FAST: "fast", SEARCH_FAST_RAYLEIGH: "fast_rayleigh"}.get(self._L.bh_engine_get_swd_search(self._h), "reference")

    @contextlib.contextmanager
    def searching(self, search):
        """The calls inside run with this root refinement; the engine's setting is restored afterwards."""
        prev = self.swd_search()
        self.set_swd_search(search)
        try:
            yield self
        finally:
            self.set_swd_search(prev)

    def set_swd_arith(self, arith):
        """Arithmetic of the launches in which every target takes the short refinement (include/bh_engine.h:
        bh_engine_set_swd_arith): "fast" (default) = fused multiply-adds, Newton-refined hardware reciprocals, polynomial
        sin / cos / exp -- same guarantees as the "fast" search; "exact" = the reference's operations and rounding points."""
        codes = {"exact": ARITH_EXACT, "fast": ARITH_FAST, ARITH_EXACT: ARITH_EXACT, ARITH_FAST: ARITH_FAST}
        code = codes.get(arith)
        if code is None:
            raise ValueError("arith must be 'exact' or 'fast'")
        self._check(self._L.bh_engine_set_swd_arith(self._h, code))

    def set_swd_trials(self, trials):
        """Trials per model and round of the trial-per-lane kernel: 0 = by the call's shape (64 up to 1024 (model, target)
        pairs, 32 up to 5120, 16 up to 10240, 8 up to 28672, 4 beyond), or 4 / 8 / 16 / 32 / 64 in every call (bh_engine_set_swd_trials)."""
        self._check(self._L.bh_engine_set_swd_trials(self._h, int(trials)))

    def swd_trials(self):
        return int(self._L.bh_engine_get_swd_trials(self._h))

    @contextlib.contextmanager
    def trying(self, trials):
        """The calls inside run the trial-per-lane kernel with this many trials per round (None: the engine's setting stays);
        the engine's setting is restored afterwards."""
        prev = self.swd_trials()
        if trials is not None:
            self.set_swd_trials(trials)
        try:
            yield self
        finally:
            if trials is not None:
                self.set_swd_trials(prev)

    def last_swd_kernel(self):
        """Which dispersion kernel the most recent call launched: "group", "lane", "lean" (bh_engine_last_swd_kernel) or None."""
        return {0: "group", 1: "lane", 2: "lean"}.get(self._L.bh_engine_last_swd_kernel(self._h))

    def swd_arith(self):
        return "fast" if self._L.bh_engine_get_swd_arith(self._h) == ARITH_FAST else "exact"

    @contextlib.contextmanager
    def computing(self, arith):
        """The calls inside run with this arithmetic; the engine's setting is restored afterwards."""
        prev = self.swd_arith()
        self.set_swd_arith(arith)
        try:
            yield self
        finally:
            self.set_swd_arith(prev)

    def set_swd_scan(self, scan):
        """Love bracket scans: "counted" = skip the steps a mode count proves to be without a sign change (same brackets, same
        bits, a third of the scan's evaluations) wherever a Love target is; "auto" (default) = only where that is measured
        to pay; "steps" = every step evaluated, as the reference does (include/bh_engine.h: bh_engine_set_swd_scan)."""
        codes = {"steps": SCAN_STEPS, "counted": SCAN_COUNTED, "auto": SCAN_AUTO}
        codes.update({v: v for v in list(codes.values())})
        code = codes.get(scan)
        if code is None:
            raise ValueError("scan must be 'steps', 'counted' or 'auto'")
        self._check(self._L.bh_engine_set_swd_scan(self._h, code))

    def swd_scan(self):
        return {SCAN_STEPS: "steps", SCAN_COUNTED: "counted"}.get(self._L.bh_engine_get_swd_scan(self._h), "auto")

    def guard_stats(self):
        """(models per target of the last dispersion call that the guard of the "fast" search ran again with the reference's
        sequence -- in the re-run launch or, one model per wavefront, in place --, re-run launches enqueued since the engine
        was created)"""
        counts = (C.c_int32 * 8)()
        n = C.c_uint64(0)
        self._check(self._L.bh_engine_guard_stats(self._h, counts, C.byref(n), None))
        return list(counts), int(n.value)

    def guard_total(self):
        """guarded (model, target) pairs since the engine was created"""
        return sum(self.guard_totals())

    def guard_totals(self):
        """the same per dispersion target of the calls (in the order the targets were registered)"""
        tot = (C.c_uint64 * 8)()
        self._check(self._L.bh_engine_guard_stats(self._h, None, None, tot))
        return [int(x) for x in tot]

    def timing_reset(self):
        self._check(self._L.bh_timing_reset(self._h))

    def timing_collect(self):
        """(ncalls, total_ms, {'swd': ms, 'rf': ms, 'like': ms}) summed over the timed calls
        since timing_reset(); waits for them to finish."""
        n = C.c_int(0)
        tot = C.c_double(0)
        fam = (C.c_double * 3)()
        self._check(self._L.bh_timing_collect(self._h, C.byref(n), C.byref(tot), fam))
        return n.value, tot.value, {"swd": fam[0], "rf": fam[1], "like": fam[2]}

    def timing_steps(self, maxn=4096):
        """Per-call times [ms] of the timed calls since timing_reset(): start of call i -> start of call i+1 (the last:
        its own span), from the engine's own events."""
        out = np.zeros(int(maxn))
        n = C.c_int(0)
        self._check(self._L.bh_timing_steps(self._h, int(maxn), out.ctypes.data_as(_d), C.byref(n)))
        return out[:n.value].copy()

    def last_timing(self):
        """(total_ms, families) of the calls since the last reset, then resets."""
        n, tot, fam = self.timing_collect()
        self.timing_reset()
        return tot, fam

    def debug_counters(self):
        out = (C.c_uint64 * 16)()
        self._check(self._L.bh_debug_counters(self._h, out))
        return [int(v) for v in out]

    def debug_trace(self):
        """Development aid: [nwaves, 4] uint64 records of the last counted dispersion launch
        (start, end in 100 MHz ticks, core cycles, rounds | wave type << 32 | HW_ID << 36)."""
        n = min(self.debug_counters()[7], 16384)
        out = (C.c_uint64 * (4 * max(n, 1)))()
        self._check(self._L.bh_debug_trace(self._h, out, n))
        return np.frombuffer(out, dtype=np.uint64).reshape(-1, 4)[:n].copy()

    def last_neval(self):
        v = C.c_uint64(0)
        self._check(self._L.bh_last_neval(self._h, C.byref(v)))
        return int(v.value)

    @staticmethod
    def _model_args(nlay, h, vp, vs, rho, layout):
        """Validate host model arrays; returns (B, Lmax, stride_l, stride_b, arrays...)."""
        h = _f64(h)
        if layout == "layer_major":   # [Lmax, B]
            Lmax, B = h.shape
            sl, sb = max(B, 1), 1
        elif layout == "model_major":  # [B, Lmax]
            B, Lmax = h.shape
            sl, sb = 1, Lmax
        else:
            raise ValueError("layout must be 'layer_major' or 'model_major'")
        arrs = [h] + [None if a is None else _f64(a) for a in (vp, vs, rho)]
        for a in arrs:
            if a is not None and a.shape != h.shape:
                raise ValueError("model arrays must have identical shapes")
        nlay = np.ascontiguousarray(nlay, dtype=np.int32)
        if nlay.shape != (B,):
            raise ValueError("nlay must have shape (B,)")
        if B and (nlay.min() < 1 or nlay.max() > Lmax):
            raise ValueError("nlay entries must be in 1..Lmax")
        return B, Lmax, sl, sb, nlay, arrs

    # -- host API ---------------------------------------------------------------------------
    def swd_batch(self, nlay, h, vp, vs, rho, periods, iwave, igr, mode=1, flsph=0,
                  layout="layer_major"):
        """Batched surfdisp96.  Returns (vel[B, K], err[B])."""
        B, Lmax, sl, sb, nlay, (h, vp, vs, rho) = self._model_args(nlay, h, vp, vs, rho, layout)
        periods = _f64(periods)
        K = periods.size
        vel = np.zeros((B, K))
        err = np.zeros(B, dtype=np.int32)
        self._check(self._L.bh_swd_batch(self._h, HOST, None, B, Lmax, _ptr(nlay), _ptr(h), _ptr(vp),
                                         _ptr(vs), _ptr(rho), sl, sb, K, _ptr(periods), int(iwave),
                                         int(igr), int(mode), int(flsph), _ptr(vel), _ptr(err)))
        return vel, err

    def rf_batch(self, nlay, h, vp, vs, rho, p, gauss, nsamp, fsamp, tshift, waveno, nkeep,
                 nsv=0.0, qp=None, qs=None, layout="layer_major"):
        """Batched rfmini synrf.  Returns rf[B, nkeep]."""
        B, Lmax, sl, sb, nlay, (h, vp, vs, rho) = self._model_args(nlay, h, vp, vs, rho, layout)
        qp = None if qp is None else _f64(qp)
        qs = None if qs is None else _f64(qs)
        rf = np.zeros((B, int(nkeep)))
        self._check(self._L.bh_rf_batch(self._h, HOST, None, B, Lmax, _ptr(nlay), _ptr(h), _ptr(vp),
                                        _ptr(vs), _ptr(rho), _ptr(qp), _ptr(qs), sl, sb, float(p),
                                        float(gauss), int(nsamp), float(fsamp), float(tshift),
                                        float(nsv), int(waveno), int(nkeep), _ptr(rf)))
        return rf

    def set_targets(self, descs):
        """descs: list of dicts with the `bh_target_desc` fields (arrays as numpy)."""
        arr = (TargetDesc * max(1, len(descs)))()
        keep = []
        ldy = 0
        for i, d in enumerate(descs):
            t = arr[i]
            t.kind, t.law, t.n = int(d["kind"]), int(d["law"]), int(d["n"])
            t.iwave, t.igr = int(d.get("iwave", 2)), int(d.get("igr", 0))
            t.mode, t.flsph = int(d.get("mode", 1)), int(d.get("flsph", 0))
            t.waveno, t.nsamp = int(d.get("waveno", 0)), int(d.get("nsamp", 0))
            t.p_s_per_deg, t.gauss = float(d.get("p", 6.4)), float(d.get("gauss", 1.0))
            t.fsamp, t.tshift, t.nsv = float(d.get("fsamp", 1.0)), float(d.get("tshift", 0.0)), float(d.get("nsv", 0.0))
            for key in ("x", "yobs", "yerr", "rinv"):
                v = d.get(key)
                if v is not None:
                    v = _f64(v)
                    keep.append(v)
                    setattr(t, key, v.ctypes.data_as(_d))
            t.logdet_r = float(d.get("logdet_r", 0.0))
            ldy += t.n
        self._check(self._L.bh_targets_set(self._h, len(descs), arr))
        self.ntargets = len(descs)
        self.ldy = ldy

    def evaluate_batch(self, nlay, h, vp, vs, noise, rho=None, layout="layer_major", want_ymod=False):
        """Batched JointTarget.evaluate.  noise[B, 2*nt].  Returns (logL[B], misfits[B, nt+1],
        err[B][, ymod[B, ldy]])."""
        B, Lmax, sl, sb, nlay, (h, vp, vs, rho) = self._model_args(nlay, h, vp, vs, rho, layout)
        nt = self.ntargets
        noise = _f64(noise)
        if noise.shape != (B, 2 * nt):
            raise ValueError("noise must have shape (B, 2*ntargets)")
        logL = np.zeros(B)
        misf = np.zeros((B, nt + 1))
        err = np.zeros(B, dtype=np.int32)
        ymod = np.zeros((B, self.ldy)) if want_ymod else None
        self._check(self._L.bh_evaluate_batch(self._h, HOST, None, B, Lmax, _ptr(nlay), _ptr(h),
                                              _ptr(vp), _ptr(vs), _ptr(rho), sl, sb, _ptr(noise),
                                              _ptr(logL), _ptr(misf), _ptr(err), _ptr(ymod)))
        return (logL, misf, err, ymod) if want_ymod else (logL, misf, err)

    def loglike_batch(self, ymod, noise, fail=None):
        """Likelihood of caller-supplied synthetics ymod[B, ldy] (see bh_loglike_batch)."""
        ymod = _f64(ymod)
        B = ymod.shape[0]
        nt = self.ntargets
        if ymod.shape != (B, self.ldy):
            raise ValueError("ymod must have shape (B, %d)" % self.ldy)
        noise = _f64(noise)
        if noise.shape != (B, 2 * nt):
            raise ValueError("noise must have shape (B, 2*ntargets)")
        if fail is not None:
            fail = np.ascontiguousarray(fail, dtype=np.int32)
            if fail.shape != (nt, B):
                raise ValueError("fail must have shape (ntargets, B)")
        logL = np.zeros(B)
        misf = np.zeros((B, nt + 1))
        err = np.zeros(B, dtype=np.int32)
        self._check(self._L.bh_loglike_batch(self._h, HOST, None, B, _ptr(ymod), _ptr(fail), _ptr(noise),
                                             _ptr(logL), _ptr(misf), _ptr(err)))
        return logL, misf, err

    def set_tuning(self, name, value):
        """An experiment switch of the library (csrc/bh_tuning.h; process-wide, never changes a result)."""
        self._check(self._L.bh_engine_set_tuning(self._h, name.encode(), int(value)))

    def tuning(self, name):
        v = C.c_int(0)
        self._check(self._L.bh_engine_get_tuning(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def probe_math(self, op, x):
        x = _f64(x).ravel()
        out = np.zeros(x.size // 2 if op in (6, 7) else x.size)
        self._check(self._L.bh_probe_math(self._h, int(op), out.size, x.ctypes.data_as(_d), out.ctypes.data_as(_d)))
        return out

    # -- device API (raw pointers; asynchronous) ---------------------------------------------
    def swd_batch_dev(self, B, Lmax, nlay, h, vp, vs, rho, sl, sb, K, periods, iwave, igr, vel, err,
                      stream=None, mode=1, flsph=0):
        self._check(self._L.bh_swd_batch(self._h, DEVICE, stream, B, Lmax, nlay, h, vp, vs, rho, sl, sb,
                                         K, periods, iwave, igr, mode, flsph, vel, err))

    def rf_batch_dev(self, B, Lmax, nlay, h, vp, vs, rho, sl, sb, p, gauss, nsamp, fsamp, tshift,
                     waveno, nkeep, rf, stream=None, nsv=0.0, qp=None, qs=None):
        self._check(self._L.bh_rf_batch(self._h, DEVICE, stream, B, Lmax, nlay, h, vp, vs, rho, qp, qs,
                                        sl, sb, p, gauss, nsamp, fsamp, tshift, nsv, waveno, nkeep, rf))

    def evaluate_batch_dev(self, B, Lmax, nlay, h, vp, vs, rho, sl, sb, noise, logL, misfits, err,
                           ymod=None, stream=None):
        self._check(self._L.bh_evaluate_batch(self._h, DEVICE, stream, B, Lmax, nlay, h, vp, vs, rho,
                                              sl, sb, noise, logL, misfits, err, ymod))


    def chain_propose(self, cfg, state, C_, iiter):
        rc = self._L.bh_chain_propose(self.stream, C.byref(cfg), C.byref(state), int(C_), int(iiter))
        if rc != BH_OK:
            raise EngineError("bh_chain_propose failed (%d)" % rc)

    def chain_accept(self, cfg, state, C_, iiter, logL, misfits):
        rc = self._L.bh_chain_accept(self.stream, C.byref(cfg), C.byref(state), int(C_), int(iiter), logL, misfits)
        if rc != BH_OK:
            raise EngineError("bh_chain_accept failed (%d)" % rc)

    def chain_propose_window(self, cfg, state, C_, iiter, depth, ld):
        rc = self._L.bh_chain_propose_window(self.stream, C.byref(cfg), C.byref(state), int(C_), int(iiter), int(depth), int(ld))
        if rc != BH_OK:
            raise EngineError("bh_chain_propose_window failed (%d)" % rc)

    def chain_accept_window(self, cfg, state, C_, iiter, depth, ld, logL, misfits):
        rc = self._L.bh_chain_accept_window(self.stream, C.byref(cfg), C.byref(state), int(C_), int(iiter), int(depth), int(ld),
                                            logL, misfits)
        if rc != BH_OK:
            raise EngineError("bh_chain_accept_window failed (%d)" % rc)


_default = {}


def default_engine(device=0):
    """Process-wide engine per device (what the plugin classes use)."""
    e = _default.get(device)
    if e is None:
        e = Engine(device)
        _default[device] = e
    return e
