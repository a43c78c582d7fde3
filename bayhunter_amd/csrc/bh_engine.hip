// bayhunter_amd/csrc/bh_engine.hip -- host side of the C ABI declared in include/bh_engine.h.
//
// Owns the device workspace, the stream and the registered target data of one GPU, stages
// host buffers when the caller asks for memspace = BH_HOST and launches the gfx950 kernels of
// swd_kernel.hip / rf_kernel.hip / like_kernel.hip.  There is no CPU code path in here: if no
// HIP device is usable, bh_engine_create fails.
#include "../../include/bh_engine_debug.h"
#include "bh_device.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "bh_tuning.h"
#include <cstring>
#include <mutex>

// ---- the experiment switches (bh_tuning.h) ----
namespace {
BhTuning g_tuning;
std::once_flag g_tuning_once;
int tuning_value(const char *env, const char *txt, int dflt)
{
    if (!txt) return dflt;
    if (!std::strcmp(env, "BH_SWD_SEARCH")) return (txt[0] == 'f' || txt[0] == '1') ? ((std::strstr(txt, "ray") || txt[0] == '2') ? BH_SEARCH_FAST_RAYLEIGH : BH_SEARCH_FAST) : (txt[0] == '2' ? BH_SEARCH_FAST_RAYLEIGH : BH_SEARCH_REFERENCE);
    if (!std::strcmp(env, "BH_SWD_ARITH")) return (txt[0] == 'e' || txt[0] == '0') ? BH_ARITH_EXACT : BH_ARITH_FAST;
    if (!std::strcmp(env, "BH_SWD_SCAN")) return (txt[0] == 's' || txt[0] == '0') ? BH_SCAN_STEPS : ((txt[0] == 'c' || txt[0] == '1') ? BH_SCAN_COUNTED : BH_SCAN_AUTO);
    char *end = nullptr;
    const long v = std::strtol(txt, &end, 10);
    if (end == txt) return 1; // a flag set to some text
    return (int)v;
}
// bounds: a switch can cost time, never memory safety (after the environment is parsed and after every bh_tuning_set)
void tuning_clamp()
{
    if (g_tuning.rf_lds_gated < 0 || g_tuning.rf_lds_gated > 160 * 1024) g_tuning.rf_lds_gated = 0;
    if (g_tuning.rf_lds_beside > 160 * 1024) g_tuning.rf_lds_beside = -1;
    if (g_tuning.swd_love_inlook < 0 || g_tuning.swd_love_inlook > 4) g_tuning.swd_love_inlook = 0;
    if (g_tuning.swd_gsplit < 0 || g_tuning.swd_gsplit > (1 << 24)) g_tuning.swd_gsplit = 0; // (the launch's entry index is an int)
}
void tuning_parse()
{
#ifndef BH_NO_EXPERIMENTS
#define X(field, env, dflt, doc) g_tuning.field = tuning_value(env, std::getenv(env), dflt);
    BH_TUNING_TABLE(X)
#undef X
    tuning_clamp();
#endif
    g_tuning.under_pmc = std::getenv("ROCPROF_COUNTER_COLLECTION") != nullptr ? 1 : 0;
}
} // namespace
const BhTuning &bh_tuning()
{
    std::call_once(g_tuning_once, tuning_parse);
    return g_tuning;
}
int bh_tuning_set(const char *name, int value)
{
#ifdef BH_NO_EXPERIMENTS
    (void)name; (void)value;
    return -1;
#else
    (void)bh_tuning();
    if (!name) return -1;
#define X(field, env, dflt, doc) if (!std::strcmp(name, #field)) { g_tuning.field = value; tuning_clamp(); return 0; }
    BH_TUNING_TABLE(X)
#undef X
    return -1;
#endif
}
int bh_tuning_get(const char *name, int *value)
{
    const BhTuning &t = bh_tuning();
    if (!name || !value) return -1;
#define X(field, env, dflt, doc) if (!std::strcmp(name, #field)) { *value = t.field; return 0; }
    BH_TUNING_TABLE(X)
#undef X
    return -1;
}

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct TargetHost {
    bh_target_desc d{};
    int off = 0; // column offset in a ymod row
    DevBuf x, yobs, yerr_scaled, rinv, quad;
    DevBuf sums;       // receiver function, fused likelihood: [B][4] sums of the last call (RfKernelArgs::sums)
    bool fused = false; // this call's likelihood of the target comes from `sums` (set per call by bh_evaluate_batch)
    DevBuf x60, vel60; // > 60 periods: the 60-point grid the forward model runs on + its output
    int kfwd = 0;      // periods the forward model computes (n, or 60 when n > 60)
    double logdet_extra = 0.0;
};

} // namespace

struct bh_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;             // receiver-function kernels run here, next to the dispersion kernel
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t aux2 = nullptr;            // second dispersion target of a lane-per-model call runs here, beside the first
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;
    int look_r = 0, look_l = 0;            // BH_SWD_LOOK_R / BH_SWD_LOOK_L env (experiment switches): trials per round by wave type
    bool overlap_rf = true;                // BH_NO_OVERLAP env turns it off (A/B testing)
    int rf_lds_beside_swd = 40 * 1024;     // BH_RF_LDS_BESIDE env (bytes; experiment switch), see launch_rf
    // co-resident receiver function (see bh_evaluate_batch): the dispersion kernel counts its started workgroups here
    unsigned *started = nullptr;           // device word (signal memory where available), cumulative over launches
    unsigned started_expected = 0;         // value after every launch enqueued so far has started
    int swd_prio_low = 1;                  // BH_SWD_PRIO_LOW env: dispersion wavefronts' low priority while RF wavefronts run beside them
    SwdPairWork pairwork{};                // SIMD-pairing order of the group kernel (bh_device.h)
    bool no_pair = false;                  // BH_SWD_NO_PAIR env: order by depth only (A/B testing)
    int swd_trials = 0;                    // bh_engine_set_swd_trials: trials per round of the trial-per-lane kernel (0 = by the call's shape)
    int last_swd_kernel = -1;              // bh_engine_last_swd_kernel
    SwdLaunchInfo last_swd{};              // of the most recent group-kernel launch (workgroups == 0: none)
    int last_swd_wpb = 2;                  // its wavefronts per workgroup
    int err_t_nt = -1, err_t_B = -1;       // layout for which err_t's untouched rows are known to be zero
    int swd_prio_low_now = 0;              // per call: what the next dispersion launch gets
    bool rf_gated_now = false;             // per call: the RF stream waits for the dispersion kernel's workgroups to be resident
    std::string err;
    // staging / workspace
    DevBuf nlay, h, vp, vs, rho, qp, qs, periods, vel, errb, rf, coef, ymod, noise, logL,
        misfits, err_t, probe_in, probe_out, counter, sph, perm, board, nevhi, rfz, gfirst, nevhi2;
    unsigned swd_stamp = 0;                // launch counter of the group kernel (marks its progress-board entries)
    // targets
    int nt = 0;
    int ldy = 0;
    std::vector<TargetHost> targets;
    // instrumentation
    bool timing = false, counting = false;
    bool no_mfma = false; // BH_NO_MFMA env: Gauss law through the in-kernel mat-vec (A/B testing)
    bool no_order = false; // BH_NO_ORDER env: wavefronts take the models in batch order (A/B testing)
    bool as_given = false; // bh_engine_set_model_order(e, 0): the caller's batches need no sorting by depth
    int force_group = 0; // BH_SWD_GROUP env / bh_engine_set_swd_group: 0 = choose automatically
    int force_look = 0;  // BH_SWD_LOOKAHEAD env / bh_engine_set_swd_lookahead: 0 = choose automatically
    int hint_layers = 0; // bh_engine_set_typical_layers: typical layer count of device-resident batches
    int swd_search = BH_SEARCH_FAST;  // bh_engine_set_swd_search / BH_SWD_SEARCH=reference|fast|fast_rayleigh: the short refinement (with its guard) for fundamental-mode phase-velocity targets unless told otherwise
    int swd_arith = BH_ARITH_FAST; // bh_engine_set_swd_arith / BH_SWD_ARITH=exact|fast: fast arithmetic in launches where every target takes the short refinement
    int swd_scan = 2;    // bh_engine_set_swd_scan / BH_SWD_SCAN=steps|counted|auto: Love scans skip the steps a mode count proves empty (same bits)
    DevBuf guard;        // short refinement: per target a count and a list of the models its guard fired on (re-run, see launch_swd_rerun)
    uint64_t rerun_launches = 0; // re-run launches enqueued so far (statistics)
    uint64_t guard_total[BH_MAX_TARGETS] = {0}; // guarded models of retired buffers (the live buffer carries its own running sum)
    bool guard_fresh = true;     // the guard buffer is new: zero its cumulative words as well
    bool guard_last = false;     // the most recent dispersion call had targets with the short refinement (its counts are in the buffer)
    int love_inlook = 0; // BH_SWD_LOVE_INLOOK env (experiment switch): Love trials inside a lane group, 0 = automatic
    // one EventSet per timed *_batch call since the last bh_timing_reset()
    struct EventSet {
        hipEvent_t ev[8];
        bool used[4];
    };
    std::vector<EventSet> evsets; // pool (events are reused after a reset)
    size_t ncalls = 0;            // sets in use
    uint64_t last_neval = 0;
    bool neval_pending = false;
};

namespace {

int fail(bh_engine *e, int code, const char *what, hipError_t he = hipSuccess)
{
    if (e) {
        e->err = what;
        if (he != hipSuccess) {
            e->err += ": ";
            e->err += hipGetErrorString(he);
        }
    }
    return code;
}

#define HIPCHK(e, call)                                                  \
    do {                                                                 \
        hipError_t _he = (call);                                         \
        if (_he != hipSuccess) return fail((e), BH_EHIP, #call, _he);    \
    } while (0)

int ensure(bh_engine *e, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return BH_OK;
    if (b.p) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (e->aux) HIPCHK(e, hipStreamSynchronize(e->aux));
        if (e->aux2) HIPCHK(e, hipStreamSynchronize(e->aux2));
        HIPCHK(e, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 4 + 256;
    hipError_t he = hipMalloc(&b.p, want);
    if (he != hipSuccess) return fail(e, BH_ENOMEM, "hipMalloc", he);
    b.cap = want;
    return BH_OK;
}

void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// number of elements spanned by a strided [Lmax][B] view (strides must be positive)
// every device buffer a registered target owns
void release_target(TargetHost &t)
{
    for (DevBuf *b : {&t.x, &t.yobs, &t.yerr_scaled, &t.rinv, &t.quad, &t.sums, &t.x60, &t.vel60}) release(*b);
}

size_t span_elems(int B, int Lmax, ptrdiff_t sl, ptrdiff_t sb)
{
    return (size_t)((ptrdiff_t)(Lmax - 1) * sl + (ptrdiff_t)(B - 1) * sb + 1);
}

__global__ void rho_from_vp_kernel(size_t n, const double *vp, double *rho)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rho[i] = vp[i] * 0.32 + 0.77; // Targets.py:319
}

int check_models(bh_engine *e, int B, int Lmax, ptrdiff_t sl, ptrdiff_t sb)
{
    if (!e) return BH_EINVAL;
    if (B < 0 || Lmax < 1 || Lmax > BH_MAX_LAYERS) return fail(e, BH_EINVAL, "bad B or Lmax (1..100)");
    if (sl <= 0 || sb <= 0) return fail(e, BH_EINVAL, "strides must be positive");
    return BH_OK;
}

struct Staged {
    const int32_t *nlay;
    const double *h, *vp, *vs, *rho, *qp, *qs;
    int typ_layers = 0; // typical layer count of the batch when the host can see it (0 = unknown)
};

// Copy the model arrays of a BH_HOST call into the engine's device buffers.
int stage_models(bh_engine *e, int B, int Lmax, ptrdiff_t sl, ptrdiff_t sb, const int32_t *nlay,
                 const double *h, const double *vp, const double *vs, const double *rho,
                 const double *qp, const double *qs, Staged &s)
{
    const size_t nel = span_elems(B, Lmax, sl, sb), bytes = nel * sizeof(double);
    int rc;
    if ((rc = ensure(e, e->nlay, (size_t)B * sizeof(int32_t)))) return rc;
    if ((rc = ensure(e, e->h, bytes))) return rc;
    if ((rc = ensure(e, e->vp, bytes))) return rc;
    if ((rc = ensure(e, e->vs, bytes))) return rc;
    if ((rc = ensure(e, e->rho, bytes))) return rc;
    HIPCHK(e, hipMemcpyAsync(e->nlay.p, nlay, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipMemcpyAsync(e->h.p, h, bytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipMemcpyAsync(e->vp.p, vp, bytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipMemcpyAsync(e->vs.p, vs, bytes, hipMemcpyHostToDevice, e->stream));
    if (rho) HIPCHK(e, hipMemcpyAsync(e->rho.p, rho, bytes, hipMemcpyHostToDevice, e->stream));
    {   // mean layer count, rounded up: what the lanes-per-model choice should be sized for
        long sum = 0;
        for (int b = 0; b < B; ++b) sum += nlay[b];
        s.typ_layers = B > 0 ? (int)((sum + B - 1) / B) : 0;
    }
    s.nlay = (const int32_t *)e->nlay.p;
    s.h = (const double *)e->h.p;
    s.vp = (const double *)e->vp.p;
    s.vs = (const double *)e->vs.p;
    s.rho = rho ? (const double *)e->rho.p : nullptr;
    s.qp = s.qs = nullptr;
    if (qp) {
        if ((rc = ensure(e, e->qp, bytes))) return rc;
        HIPCHK(e, hipMemcpyAsync(e->qp.p, qp, bytes, hipMemcpyHostToDevice, e->stream));
        s.qp = (const double *)e->qp.p;
    }
    if (qs) {
        if ((rc = ensure(e, e->qs, bytes))) return rc;
        HIPCHK(e, hipMemcpyAsync(e->qs.p, qs, bytes, hipMemcpyHostToDevice, e->stream));
        s.qs = (const double *)e->qs.p;
    }
    return BH_OK;
}

bh_engine::EventSet *cur_set(bh_engine *e)
{
    return (e->timing && e->ncalls > 0) ? &e->evsets[e->ncalls - 1] : nullptr;
}
void ev_begin(bh_engine *e, int fam, hipStream_t st)
{
    bh_engine::EventSet *s = cur_set(e);
    if (!s) return;
    if (!s->used[fam]) { // first launch of this family in the call
        (void)hipEventRecord(s->ev[2 * fam], st);
        s->used[fam] = true;
    }
}
void ev_end(bh_engine *e, int fam, hipStream_t st)
{
    bh_engine::EventSet *s = cur_set(e);
    if (s) (void)hipEventRecord(s->ev[2 * fam + 1], st);
}
void call_begin(bh_engine *e, hipStream_t st)
{
    e->neval_pending = false;
    if (!e->timing) return;
    if (e->ncalls == e->evsets.size()) {
        bh_engine::EventSet s{};
        for (auto &ev : s.ev)
            if (hipEventCreate(&ev) != hipSuccess) return; // timing silently unavailable
        e->evsets.push_back(s);
    }
    bh_engine::EventSet &s = e->evsets[e->ncalls++];
    for (bool &u : s.used) u = false;
    (void)hipEventRecord(s.ev[6], st);
    s.used[3] = true;
}
void call_end(bh_engine *e, hipStream_t st)
{
    bh_engine::EventSet *s = cur_set(e);
    if (s) (void)hipEventRecord(s->ev[7], st);
}

int swd_supported(bh_engine *e, int K, int iwave, int mode, int flsph)
{
    if (K < 0 || K > BH_MAX_PERIODS) return fail(e, BH_EINVAL, "K must be 0..60 (surfdisp96.f:61-62)");
    if (iwave != BH_WAVE_LOVE && iwave != BH_WAVE_RAYLEIGH) return fail(e, BH_EINVAL, "iwave must be 1 (Love) or 2 (Rayleigh)");
    if (mode < 1 || mode > 16) return fail(e, BH_EINVAL, "mode must be 1..16");
    if (flsph != 0 && flsph != 1) return fail(e, BH_EINVAL, "flsph must be 0 or 1");
    return BH_OK;
}

struct SwdJob {
    int K, iwave, igr, ldv;
    const double *periods_dev;
    double *vel;
    int32_t *err;
    int mode = 1;
    int flsph = 0;
    double *first = nullptr; // igr = 2 (launch_swd_jobs: the chain of a group velocity's first roots): SwdKernelArgs::first
};

int swd_counter(bh_engine *e, hipStream_t st, unsigned long long **out)
{
    *out = nullptr;
    if (!e->counting) return BH_OK;
    int rc = ensure(e, e->counter, BH_DEBUG_WORDS * sizeof(unsigned long long));
    if (rc) return rc;
    if (!e->neval_pending) {
        HIPCHK(e, hipMemsetAsync(e->counter.p, 0, BH_COUNTER_WORDS * sizeof(unsigned long long), st));
        e->neval_pending = true;
    }
    *out = (unsigned long long *)e->counter.p;
    return BH_OK;
}

// The guard of the short refinement (SearchT, swd_common.h): work space = three blocks of BH_MAX_TARGETS counts (GUARD_HEAD = 24 words, below) followed by one
// list of B model indices per target.
constexpr int GUARD_HEAD = 24;
int guard_space(bh_engine *e, hipStream_t st, int B, int32_t **counts, int32_t **lists)
{
    const size_t need = ((size_t)GUARD_HEAD + (size_t)BH_MAX_TARGETS * (size_t)(B + 4)) * sizeof(int32_t);
    if (need > e->guard.cap) { // (a new buffer: carry the cumulative counts over on the host side)
        if (e->guard.p) {
            int32_t cum[BH_MAX_TARGETS];
            HIPCHK(e, hipStreamSynchronize(st));
            HIPCHK(e, hipMemcpy(cum, (int32_t *)e->guard.p + 2 * BH_MAX_TARGETS, sizeof(cum), hipMemcpyDeviceToHost));
            for (int t = 0; t < BH_MAX_TARGETS; ++t) e->guard_total[t] += (uint64_t)(uint32_t)cum[t]; // (the device word wraps as unsigned)
        }
        e->guard_fresh = true;
    }
    int rc = ensure(e, e->guard, need);
    if (rc) return rc;
    *counts = (int32_t *)e->guard.p;
    *lists = (int32_t *)e->guard.p + GUARD_HEAD;
    // words [0, 8): this call's lists' lengths (re-run launch); [8, 16): this call's models restarted in place (group kernel,
    // one model per wavefront); [16, 24): cumulative since the buffer was made
    HIPCHK(e, hipMemsetAsync(e->guard.p, 0, (e->guard_fresh ? GUARD_HEAD : 2 * BH_MAX_TARGETS) * sizeof(int32_t), st));
    e->guard_fresh = false;
    return BH_OK;
}

// Second launch of a call in BH_SEARCH_FAST: the models the guard fired on (per target: counts[t], lists + t * (B + 4)),
// again, with the reference's sequence -- one model per wavefront, as many trials per round as its lanes admit.  The launch
// is sized for the worst case and reads the counts on the device: workgroups beyond them leave at once (no host round trip;
// nearly always all of them).  Rows and failure flags of the listed models are overwritten with the reference's.
int launch_swd_rerun(bh_engine *e, hipStream_t st, const SwdMultiArgs &main, int32_t *counts, int32_t *lists)
{
    SwdMultiArgs a = main;
    a.fast = 0;
    a.rerun = 1;
    a.restart = 0;
    a.perm = nullptr;
    a.split = nullptr;
    a.Lcut = a.Lmax;
    a.started = nullptr;
    a.stamp = (++e->swd_stamp) & 0xffffu;
    if (a.stamp == 0) a.stamp = (++e->swd_stamp) & 0xffffu;
    a.adapt_ok = 1;
    const int G = bh_swd_pick_group(a.B, a.ntargets, a.Lmax);
    for (int t = 0; t < a.ntargets; ++t) {
        a.t[t].perm = lists + (size_t)t * (size_t)(a.B + 4);
        a.t[t].count = counts + t;
        a.t[t].gcount = nullptr;
        a.t[t].glist = nullptr;
        a.t[t].look = 64 / G > 1 ? 64 / G : 1; // one model per wavefront
        a.t[t].inlook = 1;
    }
    SwdLaunchInfo info{};
    const int lrc = bh_launch_swd_group(a, G, st, &info, 2, nullptr);
    if (lrc != 0) return fail(e, BH_EINVAL, "model too deep for LDS");
    HIPCHK(e, hipGetLastError());
    ++e->rerun_launches;
    return BH_OK;
}

// All dispersion targets of one call.  Small batches go to the group kernel (G lanes per model,
// one launch for all targets); batches that fill the chip by themselves use one lane per model.
int launch_swd_jobs(bh_engine *e, hipStream_t st, int B, int Lmax, const Staged &m, ptrdiff_t sl,
                    ptrdiff_t sb, int njobs, const SwdJob *jobs_given)
{
    if (B == 0 || njobs == 0) return BH_OK;
    int rc;
    // Group velocities of the fundamental mode: two launches.  The root at t/(1+h) of period k + 1 starts from the root at
    // t/(1+h) of period k (surfdisp96.f:262-266); the root at t/(1-h) starts from the root at t/(1+h) of ITS OWN period
    // (:282-287) and nothing starts from it.  One search after the other is a chain of 2 K dependent roots; here the target
    // first runs as the chain of its K first roots (igr = 2: beside the call's other targets, in whatever kernel the call
    // takes, every root stored unrounded in the output row), then a launch of B x K independent searches (one lane each, or
    // up to 16 trial lanes while that still fits the chip at once: swd_kernel, SwdKernelArgs::second) finds the second roots
    // and puts the group velocities in their place.  The same searches on the same values: the same bits (tests/test_gpu_swd.py's
    // golden rows and oracle comparisons run through here).  Measured, one launch -> two (ms per call): one model 2.26 -> 1.43;
    // two targets x 256 / 1024 / 4096 / 8192 / 16 384 / 65 536 models: 2.64 -> 1.90, 2.91 -> 2.10, 5.51 -> 4.64, 11.2 -> 8.09,
    // 17.2 -> 12.5, 31.5 -> 29.9 (the second roots are a third of the evaluations but run as full wavefronts of independent
    // searches, one lane each; in one launch they sit in the chain's wavefronts, whose lanes wait for each other's searches).
    // bh_tuning.h swd_gsplit: the largest call in (model, period) pairs that is split (0: none).
    SwdJob jobs_split[BH_MAX_TARGETS];
    const SwdJob *jobs = jobs_given;
    int nsplit = 0;
    {
        const long cap = bh_tuning().swd_gsplit;
        bool take[BH_MAX_TARGETS] = {false};
        for (int j = 0; j < njobs && njobs <= BH_MAX_TARGETS; ++j) {
            take[j] = jobs_given[j].K != 0 && jobs_given[j].igr == 1 && jobs_given[j].mode <= 1 && cap > 0 && (long)B * jobs_given[j].K <= cap;
            nsplit += take[j] ? 1 : 0;
        }
        if (nsplit > 0) {
            if ((rc = ensure(e, e->gfirst, (size_t)nsplit * B * sizeof(double)))) return rc;
            int n = 0;
            for (int j = 0; j < njobs; ++j) {
                jobs_split[j] = jobs_given[j];
                if (take[j]) {
                    jobs_split[j].igr = 2;
                    jobs_split[j].first = (double *)e->gfirst.p + (size_t)(n++) * B;
                }
            }
            jobs = jobs_split;
        }
    }
    int kmax = 0, maxmode = 1, nlive = 0;
    bool any_sphere = false;
    for (int j = 0; j < njobs; ++j) {
        if (jobs[j].K == 0) continue;
        ++nlive;
        kmax = jobs[j].K > kmax ? jobs[j].K : kmax;
        maxmode = jobs[j].mode > maxmode ? jobs[j].mode : maxmode;
        any_sphere = any_sphere || jobs[j].flsph == 1;
    }
    if (nlive == 0) return BH_OK;
    // earth-flattened copies of the batch (layer-major), made once per call
    const double *sh = nullptr, *svp = nullptr, *svs = nullptr, *srl = nullptr, *srr = nullptr;
    if (any_sphere) {
        const size_t nb = (size_t)Lmax * B * sizeof(double);
        if ((rc = ensure(e, e->sph, 5 * nb))) return rc;
        double *base = (double *)e->sph.p;
        const size_t ne = (size_t)Lmax * B;
        bh_launch_sphere(B, Lmax, m.nlay, m.h, m.vp, m.vs, m.rho, sl, sb, base, base + ne, base + 2 * ne,
                         base + 3 * ne, base + 4 * ne, st);
        sh = base; svp = base + ne; svs = base + 2 * ne; srl = base + 3 * ne; srr = base + 4 * ne;
    }
    // Typical depth of the batch (0 = unknown: the array capacity is planned for).  Ignored where every (model, target) gets a
    // wavefront of its own anyway (at most 2048 of them: the chip's two per SIMD) -- the regime of the chains' speculative
    // windows.  There a depth class of its own for the shallow bulk buys nothing (the models per wavefront cannot grow), the
    // narrower lane groups cost a second pass over the layers and seven instead of four trials per round more transitions:
    // measured on windows of 1016 models of 3-9 layers in arrays of 21 (c4): 1.50 ms without the hint, 1.70-1.83 ms with it.
    const int typ_given = m.typ_layers > 0 ? m.typ_layers : e->hint_layers;
    const bool hint_always = bh_tuning().swd_hint_always != 0;
    const int typ_layers = ((long)nlive * B <= 2048 && !hint_always) ? 0 : typ_given;
    int iw[BH_MAX_TARGETS], look[BH_MAX_TARGETS], G = 1;
    {
        int n = 0;
        for (int j = 0; j < njobs; ++j)
            if (jobs[j].K != 0) iw[n++] = jobs[j].iwave;
        // lanes per model follow the TYPICAL depth of the batch (deeper models take further passes over
        // their layers): known for host batches, a caller's hint for device-resident ones, else Lmax
        int Lplan = typ_layers > 0 ? typ_layers : Lmax;
        if (Lplan > Lmax) Lplan = Lmax;
        bh_swd_plan(B, Lplan, n, iw, e->force_group, &G, look);
        if (e->force_look > 0)
            for (int t = 0; t < n; ++t) look[t] = e->force_look; // (one lane per model: rounded down to a power of two)
    }
    const size_t lds_cap = 64 * 1024;
    if (G <= 1 && bh_swd_lds_bytes(Lmax, kmax, maxmode) > lds_cap) G = 2; // deep models / many periods
    // The trial-per-lane kernel (swd_lean.hip): every target of the call takes the short refinement with the fast arithmetic, in
    // calls of up to 2^20 (model, target) pairs (with four trials per round it stays ahead of one lane per evaluation -- swd_kernel's
    // FA builds -- as far as measured: c2 at B = 65 536 7.99 against 8.44 ms).
    int lean_trials = 0;
    {
        bool all = e->swd_arith == BH_ARITH_FAST && e->swd_search == BH_SEARCH_FAST && e->force_group == 0 && e->force_look == 0 &&
                   e->look_r == 0 && e->look_l == 0 && bh_tuning().swd_no_lean == 0 && maxmode <= 1 && kmax <= BH_MAX_PERIODS && Lmax <= 32 &&
                   (long)B * nlive <= (bh_tuning().swd_lean_pairs > 0 ? (long)bh_tuning().swd_lean_pairs : (1L << 20));
        for (int j = 0; j < njobs; ++j) all = all && (jobs[j].K == 0 || jobs[j].igr == 0);
        if (all) lean_trials = e->swd_trials > 0 ? e->swd_trials : bh_swd_lean_trials(B, nlive);
        if (lean_trials >= 4 && bh_swd_lean_lds_bytes(lean_trials, Lmax, kmax) > lds_cap) lean_trials = 0; // (a workgroup's LDS)
    }
    const bool lean = lean_trials >= 4;
    if (lean && G <= 1) G = bh_swd_pick_group(B, nlive, Lmax); // (the launch is set up where the group kernel's is)
    unsigned long long *counter = nullptr;
    if ((rc = swd_counter(e, st, &counter))) return rc;
    // the launches of the second roots of the group-velocity targets run as two chains (above), after the call's other launches
    auto second_roots = [&]() -> int {
        int J2 = 16; // trial lanes per search while all searches of the call still fit the chip at once (2048 wavefronts)
        {
            long searches = 0;
            for (int j = 0; j < njobs; ++j) searches += jobs[j].igr == 2 ? (long)B * jobs[j].K : 0;
            while (J2 > 1 && searches * J2 > 2048L * 64) J2 >>= 1;
        }
        size_t off[BH_MAX_TARGETS + 1] = {0};
        int n = 0;
        for (int j = 0; j < njobs; ++j)
            if (jobs[j].igr == 2) {
                off[n + 1] = off[n] + bh_swd_nev_high_doubles(B * jobs[j].K, J2);
                ++n;
            }
        int rc2 = ensure(e, e->nevhi2, off[n] * sizeof(double));
        if (rc2) return rc2;
        // (as with the one-lane-per-model launches below: every second target on a second stream, side by side)
        const bool fork = n > 1 && e->aux2 != nullptr;
        if (fork) {
            HIPCHK(e, hipEventRecord(e->ev_fork2, st));
            HIPCHK(e, hipStreamWaitEvent(e->aux2, e->ev_fork2, 0));
        }
        n = 0;
        for (int j = 0; j < njobs; ++j) {
            const SwdJob &J = jobs[j];
            if (J.igr != 2) continue;
            SwdKernelArgs a{};
            a.B = B * J.K; a.Bm = B; a.second = 1; a.first = J.first;
            a.Lmax = Lmax; a.K = J.K; a.igr = 1; a.mode = 1; a.perm = nullptr;
            a.nlay = m.nlay; a.h = m.h; a.vp = m.vp; a.vs = m.vs; a.rho = m.rho; a.sl = sl; a.sb = sb;
            if (J.flsph == 1) {
                a.h = sh; a.vp = svp; a.vs = svs; a.rho = (J.iwave == BH_WAVE_LOVE) ? srl : srr;
                a.sl = B; a.sb = 1;
            }
            a.periods = J.periods_dev; a.vel = J.vel; a.ldv = J.ldv; a.err = J.err; a.neval = counter;
            a.look = J2;
            const long waves = ((long)a.B * J2 + 63) / 64;
            a.fair = waves <= 1024 ? -1 : (waves <= 2048 ? 18 : 12);
            a.nev_high = (double *)e->nevhi2.p + off[n];
            a.fast = 0; a.farith = 0; a.counted = e->swd_scan;
            bh_launch_swd(a, J.iwave, (fork && (n & 1)) ? e->aux2 : st);
            ++n;
        }
        if (fork) {
            HIPCHK(e, hipEventRecord(e->ev_join2, e->aux2));
            HIPCHK(e, hipStreamWaitEvent(st, e->ev_join2, 0));
        }
        HIPCHK(e, hipGetLastError());
        return BH_OK;
    };
    // processing order: deepest models first, wavefronts of (nearly) one depth
    const int32_t *perm = nullptr, *split = nullptr;
    int Lcut = Lmax;
    // One depth class, lanes per model > 1: the launcher orders the models itself, by predicted search length and paired
    // over the SIMDs (SwdPairWork); mixed depths are ordered by depth there as well.
    // Measured (bench.py c2, B = 2048 ... 4096, same session with / without): -3 % at 4096 and -4 % at 3800, where the
    // launch nearly fills two wavefronts per SIMD and the Rayleigh wavefronts (look-ahead 2) outnumber the Love ones
    // (trials inside the lane groups) 7 : 3; neutral at 3500; +2.5 % at 2048 / 3072, where the planner gives both targets
    // the same look-ahead and every SIMD holds one Rayleigh and one Love wavefront (why the one-to-one mix loses is not
    // understood).  Hence: two targets, >= 7/8 of two wavefronts per SIMD, unequal wavefront counts.
    long plan_waves = 0, wmin = 0, wmax = 0;
    if (G > 1)
        for (int t = 0; t < nlive; ++t) {
            int J = look[t] > 1 ? look[t] : 1;
            while (J > 1 && G * J > 64) --J;
            const int mpw = 64 / (G * J);
            const long w = (B + mpw - 1) / mpw;
            plan_waves += w;
            wmin = (t == 0 || w < wmin) ? w : wmin;
            wmax = w > wmax ? w : wmax;
        }
    const int pair_min = bh_tuning().swd_pair_minwaves;
    const bool use_pair = !lean && B > 1 && !e->no_order && !e->as_given && !e->no_pair && G > 1 && nlive == 2 && bh_pair_order_fits(B) &&
                          plan_waves >= (pair_min >= 0 ? pair_min : 7 * (long)e->pairwork.ncu) && (pair_min >= 0 || 2 * wmax >= 3 * wmin) &&
                          !(typ_layers > 0 && typ_layers + 2 < Lmax);
    if (B > 1 && !e->no_order && !e->as_given && !use_pair) {
        if ((rc = ensure(e, e->perm, ((size_t)B + 4) * sizeof(int32_t)))) return rc;
        int32_t *p = (int32_t *)e->perm.p;
        // LDS rows for the bulk of the batch: its typical depth plus a margin; deeper models get their own launch
        const int typ = typ_layers;
        if (lean && bh_pair_order_fits(B) && bh_tuning().swd_lean_no_sort == 0) {
            // The trial-per-lane kernel: the models in the order of their PREDICTED search length, longest first (mixed depths:
            // deepest first) -- wavefronts of like models idle fewer rounds on finished ones, and the long wavefronts are
            // dispatched first with the short ones filling the launch's tail (c2: 0.79 -> 0.72 ms).  Scheduling only.
            // Sorted inside eight blocks of the batch, block x for the wavefronts of XCD x, where the shapes divide: each XCD's L2
            // then fetches an eighth of the model arrays instead of all of them (HBM reads of the launch 10.7 -> 1.9 MB at c2).
            PairOrderTarget tg{1, B, nullptr, p + 4, 0};
            const int mpw = BH_WAVE / lean_trials, per_wg = (nlive == 2) ? 2 : 4;
            if (bh_tuning().swd_lean_xcd != 0 && bh_tuning().swd_lean_r < 4 && bh_tuning().swd_lean_l < 4 && B % mpw == 0 && B % 8 == 0 &&
                (B / mpw) % (8 * per_wg) == 0)
                tg = PairOrderTarget{mpw, B / mpw, nullptr, p + 4, per_wg};
            bh_launch_pair_order(B, Lmax, m.nlay, m.vs, sl, sb, 1, &tg, st);
            perm = p + 4;
        } else {
            if (typ > 0 && typ + 2 < Lmax) Lcut = typ + 2;
            bh_launch_order(B, m.nlay, p + 4, Lcut, p, st);
            perm = p + 4;
            split = (Lcut < Lmax) ? p : nullptr;
        }
    }
    if (G <= 1) {
        // One lane per model: a launch per target.  A wavefront of these kernels keeps its SIMD's vector issue ~60 %
        // busy (profiles/), so the targets of a call run SIDE BY SIDE: every second one on a second stream.
        int nlive2 = 0;
        for (int j = 0; j < njobs; ++j) nlive2 += (jobs[j].K != 0);
        const bool fork2 = nlive2 > 1 && e->aux2 != nullptr;
        // work array of the launches (the Neville orders the kernel does not keep in LDS): one region per target,
        // the targets run concurrently
        size_t nev_off[BH_MAX_TARGETS + 1] = {0};
        for (int t = 0; t < nlive2; ++t) nev_off[t + 1] = nev_off[t] + bh_swd_nev_high_doubles(B, look[t] > 1 ? look[t] : 1);
        if ((rc = ensure(e, e->nevhi, nev_off[nlive2] * sizeof(double)))) return rc;
        long lane_waves = 0; // wavefronts of the call
        for (int t = 0; t < nlive2; ++t) lane_waves += (long)((B + 63) / 64) * (look[t] > 1 ? look[t] : 1);
        int32_t *gcounts = nullptr, *glists = nullptr;
        bool any_fast = false;
        for (int j = 0; j < njobs; ++j)
            any_fast = any_fast || (jobs[j].K != 0 && jobs[j].igr == 0 && jobs[j].mode <= 1 &&
                                    (e->swd_search == BH_SEARCH_FAST || (e->swd_search == BH_SEARCH_FAST_RAYLEIGH && jobs[j].iwave == BH_WAVE_RAYLEIGH)));
        e->guard_last = any_fast;
        if (any_fast && (rc = guard_space(e, st, B, &gcounts, &glists))) return rc;
        SwdMultiArgs ra{}; // (the re-run of guarded models goes through the group kernel)
        ra.B = B; ra.Lmax = Lmax; ra.nlay = m.nlay; ra.neval = counter; ra.counted = e->swd_scan;
        if (any_fast) {
            if (!e->board.p) {
                if ((rc = ensure(e, e->board, (size_t)BH_BOARD_WORDS * sizeof(unsigned)))) return rc;
                HIPCHK(e, hipMemsetAsync(e->board.p, 0, (size_t)BH_BOARD_WORDS * sizeof(unsigned), st));
            }
            ra.board = (unsigned *)e->board.p;
        }
        e->last_swd_kernel = BH_KERNEL_LANE;
        ev_begin(e, 0, st);
        if (fork2) {
            HIPCHK(e, hipEventRecord(e->ev_fork2, st));
            HIPCHK(e, hipStreamWaitEvent(e->aux2, e->ev_fork2, 0));
        }
        int nth = 0;
        for (int j = 0; j < njobs; ++j) {
            const SwdJob &J = jobs[j];
            if (J.K == 0) continue;
            SwdKernelArgs a{};
            a.B = B; a.Lmax = Lmax; a.K = J.K; a.igr = J.igr; a.mode = J.mode; a.perm = perm;
            a.nlay = m.nlay; a.h = m.h; a.vp = m.vp; a.vs = m.vs; a.rho = m.rho; a.sl = sl; a.sb = sb;
            if (J.flsph == 1) {
                a.h = sh; a.vp = svp; a.vs = svs; a.rho = (J.iwave == BH_WAVE_LOVE) ? srl : srr;
                a.sl = B; a.sb = 1;
            }
            a.periods = J.periods_dev; a.vel = J.vel; a.ldv = J.ldv; a.err = J.err; a.neval = counter;
            a.first = J.first;
            a.look = look[nth] > 1 ? look[nth] : 1;
            // priority time slice (log2 cycles) of wavefronts that share a SIMD, see swd_kernel; wavefronts with a SIMD of
            // their own are left alone (a low-priority phase costs them 8 %: the CU's front end is shared)
            a.fair = lane_waves <= 1024 ? -1 : (lane_waves <= 2048 ? 18 : 12);
            a.nev_high = (double *)e->nevhi.p + nev_off[nth];
            a.fast = (J.igr == 0 && J.mode <= 1 && (e->swd_search == BH_SEARCH_FAST || (e->swd_search == BH_SEARCH_FAST_RAYLEIGH && J.iwave == BH_WAVE_RAYLEIGH))) ? 1 : 0;
            a.counted = e->swd_scan;
            a.farith = (a.fast && e->swd_arith == BH_ARITH_FAST) ? 1 : 0;
            if (a.fast) {
                a.gcount = gcounts + nth;
                a.glist = glists + (size_t)nth * (size_t)(B + 4);
            }
            bh_launch_swd(a, J.iwave, (fork2 && (nth & 1)) ? e->aux2 : st);
            SwdTarget &t = ra.t[ra.ntargets++];
            t.iwave = J.iwave; t.igr = J.igr; t.K = J.K; t.ldv = J.ldv; t.mode = J.mode;
            t.h = a.h; t.vp = a.vp; t.vs = a.vs; t.rho = a.rho; t.sl = a.sl; t.sb = a.sb;
            t.periods = J.periods_dev; t.vel = J.vel; t.err = J.err;
            ++nth;
        }
        if (fork2) {
            HIPCHK(e, hipEventRecord(e->ev_join2, e->aux2));
            HIPCHK(e, hipStreamWaitEvent(st, e->ev_join2, 0));
        }
        HIPCHK(e, hipGetLastError());
        if (any_fast && (rc = launch_swd_rerun(e, st, ra, gcounts, glists))) return rc;
        if (nsplit > 0 && (rc = second_roots())) return rc;
        ev_end(e, 0, st);
        return BH_OK;
    }
    SwdMultiArgs a{};
    if (!e->board.p) { // progress board of the group kernel (scheduling aid; zeroed once, entries carry a launch stamp)
        if ((rc = ensure(e, e->board, (size_t)BH_BOARD_WORDS * sizeof(unsigned)))) return rc;
        HIPCHK(e, hipMemsetAsync(e->board.p, 0, (size_t)BH_BOARD_WORDS * sizeof(unsigned), st));
    }
    a.board = (unsigned *)e->board.p;
    a.stamp = (++e->swd_stamp) & 0xffffu;
    if (a.stamp == 0) a.stamp = (++e->swd_stamp) & 0xffffu;
    a.B = B; a.Lmax = Lmax; a.ntargets = 0; a.nlay = m.nlay; a.neval = counter; a.perm = perm;
    a.split = split; a.Lcut = Lcut;
    for (int j = 0; j < njobs; ++j) {
        const SwdJob &J = jobs[j];
        if (J.K == 0) continue;
        SwdTarget &t = a.t[a.ntargets++];
        t.iwave = J.iwave; t.igr = J.igr; t.K = J.K; t.ldv = J.ldv; t.mode = J.mode;
        t.look = look[a.ntargets - 1];
        // Love: two trials inside each lane group pay while the group count is small (measured: wavefront
        // 7 % shorter at look = 1, neutral at 2, slower beyond)
        t.inlook = (J.iwave != BH_WAVE_LOVE) ? 1 : (e->love_inlook > 0 ? e->love_inlook : (t.look <= 2 ? 2 : 1));
        t.h = m.h; t.vp = m.vp; t.vs = m.vs; t.rho = m.rho; t.sl = sl; t.sb = sb;
        if (J.flsph == 1) {
            t.h = sh; t.vp = svp; t.vs = svs; t.rho = (J.iwave == BH_WAVE_LOVE) ? srl : srr;
            t.sl = B; t.sb = 1;
        }
        t.periods = J.periods_dev; t.vel = J.vel; t.err = J.err;
        t.first = J.first;
    }
    // which targets take the short refinement: phase velocities; with BH_SEARCH_FAST_RAYLEIGH only the Rayleigh ones
    auto takes_fast = [&](const SwdTarget &t) {
        return t.igr == 0 && t.mode <= 1 && (e->swd_search == BH_SEARCH_FAST || (e->swd_search == BH_SEARCH_FAST_RAYLEIGH && t.iwave == BH_WAVE_RAYLEIGH));
    };
    a.counted = e->swd_scan;
    a.farith = e->swd_arith == BH_ARITH_FAST ? 1 : 0;
    for (int t = 0; t < a.ntargets; ++t) {
        if (e->look_r > 0 && a.t[t].iwave == BH_WAVE_RAYLEIGH) a.t[t].look = e->look_r;
        if (e->look_l > 0 && a.t[t].iwave == BH_WAVE_LOVE) a.t[t].look = e->look_l;
    }
    a.started = e->started;
    a.prio_low = e->swd_prio_low_now;
    a.adapt_ok = (e->force_group == 0 && e->force_look == 0 && e->look_r == 0 && e->look_l == 0) ? 1 : 0;
    {
        const bool dbg = bh_tuning().debug_plan != 0;
        if (dbg) {
            std::fprintf(stderr, "[bh] dispersion plan B=%d Lmax=%d: lanes per model %d, typical layers %d (hint %d), Lcut %d%s, pairing %d; trials per round:",
                         B, Lmax, G, m.typ_layers, e->hint_layers, Lcut, split ? " (two depth classes)" : "", (int)use_pair);
            for (int t = 0; t < a.ntargets; ++t) std::fprintf(stderr, " %s %d/%d", a.t[t].iwave == BH_WAVE_LOVE ? "L" : "R", a.t[t].look, a.t[t].inlook);
            std::fprintf(stderr, "\n");
        }
    }
    // ONE launch for all targets.  Where some phase-velocity targets take the short refinement and others keep the reference's
    // sequence (BH_SEARCH_FAST_RAYLEIGH with Love targets in the call) the launch takes the build with both sequences and the
    // targets say which is theirs (two launches side by side were measured: the Rayleigh / Love pairs on the SIMDs are lost,
    // c4 1.37 -> 1.58 ms per window).
    {
        int nfast = 0;
        for (int t = 0; t < a.ntargets; ++t) nfast += takes_fast(a.t[t]) ? 1 : 0;
        for (int t = 0; t < a.ntargets; ++t) a.t[t].refseq = (nfast > 0 && a.t[t].igr == 0 && !takes_fast(a.t[t])) ? 1 : 0;
        a.fast = nfast > 0 ? 1 : 0;
        a.restart = 1; // (takes effect in launches of one model per wavefront: see bh_launch_swd_group)
    }
    int32_t *gcounts = nullptr, *glists = nullptr;
    e->guard_last = a.fast != 0;
    if (a.fast) {
        if ((rc = guard_space(e, st, B, &gcounts, &glists))) return rc;
        for (int t = 0; t < a.ntargets; ++t)
            if (a.t[t].igr == 0) {
                a.t[t].gcount = gcounts + t;
                a.t[t].glist = glists + (size_t)t * (size_t)(B + 4);
            }
    }
    ev_begin(e, 0, st);
    int lrc;
    if (lean) {
        const BhTuning &tun = bh_tuning();
        for (int t = 0; t < a.ntargets; ++t) {
            int Jt = lean_trials;
            if (a.t[t].iwave == BH_WAVE_RAYLEIGH && tun.swd_lean_r >= 4) Jt = tun.swd_lean_r;
            if (a.t[t].iwave == BH_WAVE_LOVE && tun.swd_lean_l >= 4) Jt = tun.swd_lean_l;
            a.t[t].look = Jt;
        }
        lrc = bh_launch_swd_lean(a, st, &e->last_swd);
        e->last_swd_kernel = BH_KERNEL_LEAN;
    } else {
        e->last_swd_kernel = BH_KERNEL_GROUP;
        lrc = bh_launch_swd_group(a, G, st, &e->last_swd, 2, use_pair ? &e->pairwork : nullptr);
    }
    e->last_swd_wpb = lean ? 4 : 2; // (wavefronts per workgroup of the kernel that was launched)
    if (lrc != 0) {
        ev_end(e, 0, st);
        return fail(e, BH_EINVAL, "model too deep for LDS");
    }
    // (the counter a second stream waits on moves only once the launch is known to have been accepted: a failed launch
    // never increments the device word, and every later wait for the advanced value would hang -- ADVICE r03)
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        ev_end(e, 0, st);
        return fail(e, BH_EHIP, "dispersion kernel launch", le);
    }
    if (e->started) e->started_expected += e->last_swd.workgroups;
    if (a.fast && !e->last_swd.restarts_in_place && (rc = launch_swd_rerun(e, st, a, gcounts, glists))) {
        ev_end(e, 0, st);
        return rc;
    }
    if (nsplit > 0 && (rc = second_roots())) {
        ev_end(e, 0, st);
        return rc;
    }
    ev_end(e, 0, st);
    return BH_OK;
}

int launch_rf(bh_engine *e, hipStream_t st, int B, int Lmax, const Staged &m, ptrdiff_t sl,
              ptrdiff_t sb, double p, double gauss, int nsamp, double fsamp, double tshift,
              double nsv, int waveno, int nkeep, double *rf, int ldr, bool beside_swd = false, const double *yobs = nullptr,
              double *sums = nullptr)
{
    if (B == 0) return BH_OK;
    int rc;
    if ((rc = ensure(e, e->coef, (size_t)B * bh_rf_coef_doubles(Lmax) * sizeof(double)))) return rc;
    RfKernelArgs a{};
    a.B = B; a.Lmax = Lmax; a.nsamp = nsamp; a.nkeep = nkeep; a.waveno = waveno;
    a.nlay = m.nlay; a.h = m.h; a.vp = m.vp; a.vs = m.vs; a.rho = m.rho; a.qp = m.qp; a.qs = m.qs;
    a.sl = sl; a.sb = sb;
    a.p_s_per_deg = p; a.gauss = gauss; a.fsamp = fsamp; a.tshift = tshift; a.nsv = nsv;
    a.coef = (double *)e->coef.p; a.rf = rf; a.ldr = ldr;
    a.yobs = yobs; a.sums = sums; // (fused likelihood: the sums instead of the trace)
    if (bh_rf_lds_bytes(nsamp) > BH_RF_MAX_LDS) { // a trace longer than a workgroup's LDS holds: the spectra go through a workspace
        if ((rc = ensure(e, e->rfz, (size_t)B * (size_t)(nsamp / 2) * 2 * sizeof(double)))) return rc;
        a.zwork = (double *)e->rfz.p;
    }
    // Beside a dispersion launch (fused call, second stream): the synthesis workgroups must not take wave slots before
    // the dispersion kernel's wavefronts are resident -- that kernel counts on all of them being co-resident (one round
    // of wavefronts; displaced ones wait for a whole lifetime: 3.6 -> 7 ms measured when the 17.7 KB workgroups of
    // round 3 slipped into the 18 KB of LDS the dispersion wavefronts leave free on a CU).  Asking for more LDS than
    // that remainder keeps them out of CUs whose dispersion wavefronts are still running, as in round 2 (40 KB).
    // (with the start gate of bh_evaluate_batch in force the LDS floor is not needed: RF workgroups are dispatched
    // after every dispersion wavefront is resident and only take what finished wavefronts have freed)
    a.lds_min = (beside_swd && !e->rf_gated_now) ? e->rf_lds_beside_swd : 0;
    {   // (bh_tuning.h: LDS floor of the synthesis workgroups of a GATED fused call -- fewer of them per CU at a time)
        const int gated_floor = bh_tuning().rf_lds_gated;
        if (beside_swd && e->rf_gated_now && gated_floor > 0) a.lds_min = gated_floor;
    }
    a.coef_small = (beside_swd && e->rf_gated_now && bh_tuning().rf_coef_big == 0) ? 1 : 0;
    ev_begin(e, 1, st);
    const int lrc = bh_launch_rf(a, st);
    ev_end(e, 1, st);
    if (lrc != 0) return fail(e, BH_EUNSUPPORTED, "receiver function: nsamp above 262144 is not supported");
    HIPCHK(e, hipGetLastError());
    return BH_OK;
}

// Fill the likelihood descriptor of target t; for the Gauss law run the MFMA contraction first.
int prepare_like_target(bh_engine *e, hipStream_t st, int B, int ldy, const double *ymod_d, TargetHost &T,
                        LikeTargetDev &L)
{
    L.law = T.d.law; L.n = T.d.n; L.off = T.off;
    L.yobs = (const double *)T.yobs.p;
    L.yerr_scaled = (const double *)T.yerr_scaled.p;
    L.rinv = (const double *)T.rinv.p;
    L.logdet_extra = T.logdet_extra;
    L.quad = nullptr;
    L.nsplit = 0;
    L.pre = T.fused ? (const double *)T.sums.p : nullptr;
    if (T.d.law == BH_LAW_GAUSS && !e->no_mfma) {
        const int nsplit = bh_gauss_nsplit(B, T.d.n);
        int rc = ensure(e, T.quad, (size_t)B * nsplit * sizeof(double));
        if (rc) return rc;
        ev_begin(e, 2, st);
        bh_launch_gauss_quad(B, T.d.n, ldy, ymod_d + T.off, L.yobs, L.rinv, nsplit, (double *)T.quad.p, st);
        L.quad = (const double *)T.quad.p;
        L.nsplit = nsplit;
    }
    return BH_OK;
}

int rf_args_ok(bh_engine *e, int nsamp, int nkeep, double gauss, double fsamp, int waveno)
{
    if (nsamp < 4 || (nsamp & (nsamp - 1)) != 0)
        return fail(e, BH_EINVAL, "nsamp must be a power of two >= 4");
    // rfmini_modrf.py:62 derives nsamp = 2^ceil(log2(2 ndata)) with no upper bound; here one workgroup holds a model's
    // half-length complex spectrum in LDS up to 16384 samples and in an HBM workspace beyond, up to 2^18 samples
    if (nsamp > BH_RF_MAX_NSAMP)
        return fail(e, BH_EUNSUPPORTED, "nsamp above 262144 is not supported");
    if (nkeep < 0 || nkeep > nsamp) return fail(e, BH_EINVAL, "nkeep must be 0..nsamp");
    if (!(gauss > 0.0) || !(fsamp > 0.0)) return fail(e, BH_EINVAL, "gauss and fsamp must be > 0");
    if (waveno != BH_RF_P && waveno != BH_RF_SV) return fail(e, BH_EINVAL, "waveno must be 0 (P) or 1 (SV)");
    return BH_OK;
}

} // namespace

extern "C" {

int bh_abi_version(void) { return BH_ABI_VERSION; }

int bh_engine_create(int device, bh_engine **out)
{
    if (!out) return BH_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BH_EHIP;
    if (device < 0 || device >= ndev) return BH_EINVAL;
    bh_engine *e = new (std::nothrow) bh_engine;
    if (!e) return BH_ENOMEM;
    e->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&e->stream) != hipSuccess ||
        hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&e->aux2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_fork2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_join2, hipEventDisableTiming) != hipSuccess) {
        delete e;
        return BH_EHIP;
    }
    const BhTuning &tun = bh_tuning(); // (the experiment switches: parsed once per process, bh_tuning.h)
    if (tun.no_overlap) e->overlap_rf = false;
    if (tun.rf_lds_beside >= 0) e->rf_lds_beside_swd = tun.rf_lds_beside;
    if (tun.swd_no_pair) e->no_pair = true;
    {
        hipDeviceProp_t prop;
        e->pairwork.ncu = (hipGetDeviceProperties(&prop, device) == hipSuccess) ? prop.multiProcessorCount : 0;
    }
    if (tun.swd_prio_low >= 0) e->swd_prio_low = tun.swd_prio_low != 0 ? 1 : 0;
    {   // the counter the dispersion kernel's workgroups bump at start; hipStreamWaitValue32 polls it from the RF stream
        int can = 0;
        (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device);
        if (tun.no_started) can = 0;
        // rocprofv3 --pmc (counter collection) runs the dispatches of ALL queues one at a time; the RF stream's wait
        // packet then never sees the dispersion kernel start (measured: bench.py --workload c3 hangs under --pmc, runs
        // under --kernel-trace).  The gate is a scheduling aid only: off in that mode.
        if (tun.under_pmc) can = 0;
        // plain device memory: the wait packet polls it just as well, and atomics on signal memory
        // (hipMallocSignalMemory) cost the dispersion kernel 1 ms per launch (976 workgroups, one atomic each)
        if (can && hipMalloc((void **)&e->started, 8) != hipSuccess) e->started = nullptr;
        if (e->started && hipMemset(e->started, 0, 8) != hipSuccess) {
            (void)hipFree(e->started);
            e->started = nullptr;
        }
        (void)hipGetLastError();
    }
    e->force_group = tun.swd_group;
    e->force_look = tun.swd_lookahead;
    if (tun.swd_search >= 0) e->swd_search = tun.swd_search;
    if (tun.swd_arith >= 0) e->swd_arith = tun.swd_arith != 0 ? BH_ARITH_FAST : BH_ARITH_EXACT;
    if (tun.swd_scan >= 0) e->swd_scan = tun.swd_scan;
    e->love_inlook = tun.swd_love_inlook;
    e->look_r = tun.swd_look_r;
    e->look_l = tun.swd_look_l;
    if (tun.no_mfma) e->no_mfma = true;
    if (tun.no_order) e->no_order = true;
    *out = e;
    return BH_OK;
}

int bh_engine_set_typical_layers(bh_engine *e, int nlay)
{
    if (!e) return BH_EINVAL;
    if (nlay < 0 || nlay > BH_MAX_LAYERS) return fail(e, BH_EINVAL, "typical layer count must be 0 (unknown) or 1..100");
    e->hint_layers = nlay;
    return BH_OK;
}

int bh_engine_set_model_order(bh_engine *e, int sort_by_depth)
{
    if (!e) return BH_EINVAL;
    e->as_given = (sort_by_depth == 0);
    return BH_OK;
}

int bh_engine_set_swd_search(bh_engine *e, int search)
{
    if (!e) return BH_EINVAL;
    if (search != BH_SEARCH_REFERENCE && search != BH_SEARCH_FAST && search != BH_SEARCH_FAST_RAYLEIGH)
        return fail(e, BH_EINVAL, "search must be BH_SEARCH_REFERENCE (0), BH_SEARCH_FAST (1) or BH_SEARCH_FAST_RAYLEIGH (2)");
    e->swd_search = search;
    return BH_OK;
}

int bh_engine_get_swd_search(const bh_engine *e) { return e ? e->swd_search : 0; }

int bh_engine_set_swd_scan(bh_engine *e, int scan)
{
    if (!e) return BH_EINVAL;
    if (scan != BH_SCAN_STEPS && scan != BH_SCAN_COUNTED && scan != BH_SCAN_AUTO) return fail(e, BH_EINVAL, "scan must be BH_SCAN_STEPS, BH_SCAN_COUNTED or BH_SCAN_AUTO");
    e->swd_scan = scan;
    return BH_OK;
}
int bh_engine_get_swd_scan(const bh_engine *e) { return e ? e->swd_scan : 0; }

int bh_engine_set_tuning(bh_engine *e, const char *name, int value)
{
    if (!e) return BH_EINVAL;
    if (bh_tuning_set(name, value) != 0) return fail(e, BH_EINVAL, "unknown experiment switch (csrc/bh_tuning.h), or a build with BH_NO_EXPERIMENTS");
    return BH_OK;
}
int bh_engine_get_tuning(bh_engine *e, const char *name, int *value)
{
    if (!e) return BH_EINVAL;
    if (bh_tuning_get(name, value) != 0) return fail(e, BH_EINVAL, "unknown experiment switch (csrc/bh_tuning.h)");
    return BH_OK;
}

int bh_engine_set_swd_arith(bh_engine *e, int arith)
{
    if (!e) return BH_EINVAL;
    if (arith != BH_ARITH_EXACT && arith != BH_ARITH_FAST) return fail(e, BH_EINVAL, "arith must be BH_ARITH_EXACT (0) or BH_ARITH_FAST (1)");
    e->swd_arith = arith;
    return BH_OK;
}
int bh_engine_get_swd_arith(const bh_engine *e) { return e ? e->swd_arith : 0; }
int bh_engine_last_swd_kernel(const bh_engine *e) { return e ? e->last_swd_kernel : -1; }
int bh_engine_set_swd_trials(bh_engine *e, int trials)
{
    if (!e) return BH_EINVAL;
    if (trials != 0 && trials != 4 && trials != 8 && trials != 16 && trials != 32 && trials != 64)
        return fail(e, BH_EINVAL, "trials must be 0 (by the call's shape), 4, 8, 16, 32 or 64");
    e->swd_trials = trials;
    return BH_OK;
}
int bh_engine_get_swd_trials(const bh_engine *e) { return e ? e->swd_trials : 0; }
int bh_engine_guard_stats(bh_engine *e, int32_t *counts, uint64_t *rerun_launches, uint64_t *total)
{
    if (!e) return BH_EINVAL;
    if (rerun_launches) *rerun_launches = e->rerun_launches;
    if (counts || total) {
        int32_t w[3 * BH_MAX_TARGETS] = {0};
        if (e->guard.p && !e->guard_fresh) {
            HIPCHK(e, hipDeviceSynchronize()); // (the last call may have run on a caller's stream)
            HIPCHK(e, hipMemcpy(w, e->guard.p, sizeof(w), hipMemcpyDeviceToHost));
        }
        if (counts)
            for (int t = 0; t < BH_MAX_TARGETS; ++t) counts[t] = e->guard_last ? w[t] + w[BH_MAX_TARGETS + t] : 0;
        if (total)
            for (int t = 0; t < BH_MAX_TARGETS; ++t) total[t] = e->guard_total[t] + (uint64_t)(uint32_t)w[2 * BH_MAX_TARGETS + t];
    }
    return BH_OK;
}

int bh_engine_set_swd_lookahead(bh_engine *e, int trials_per_round)
{
    if (!e) return BH_EINVAL;
    if (trials_per_round < 0 || trials_per_round > 16)
        return fail(e, BH_EINVAL, "look-ahead must be 0 (auto) or 1..16 trial velocities per round");
    e->force_look = trials_per_round;
    return BH_OK;
}

int bh_engine_set_swd_group(bh_engine *e, int lanes_per_model)
{
    if (!e) return BH_EINVAL;
    if (lanes_per_model < 0 || lanes_per_model > 32)
        return fail(e, BH_EINVAL, "lanes per model must be 0 (auto) or 1..32");
    e->force_group = lanes_per_model;
    return BH_OK;
}

void bh_engine_destroy(bh_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (DevBuf *b : {&e->nlay, &e->h, &e->vp, &e->vs, &e->rho, &e->qp, &e->qs, &e->periods, &e->vel,
                      &e->errb, &e->rf, &e->coef, &e->ymod, &e->noise, &e->logL, &e->misfits,
                      &e->err_t, &e->probe_in, &e->probe_out, &e->counter, &e->sph, &e->perm, &e->board, &e->nevhi, &e->guard, &e->rfz, &e->gfirst, &e->nevhi2})
        release(*b);
    for (auto &t : e->targets) {
        release_target(t);
    }
    for (auto &s : e->evsets)
        for (auto &ev : s.ev)
            if (ev) (void)hipEventDestroy(ev);
    if (e->started) (void)hipFree(e->started);
    for (int t = 0; t < 2; ++t) {
        if (e->pairwork.perm[t]) (void)hipFree(e->pairwork.perm[t]);
        if (e->pairwork.slot_rank[t]) (void)hipFree(e->pairwork.slot_rank[t]);
    }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->aux) (void)hipStreamDestroy(e->aux);
    if (e->ev_fork2) (void)hipEventDestroy(e->ev_fork2);
    if (e->ev_join2) (void)hipEventDestroy(e->ev_join2);
    if (e->aux2) (void)hipStreamDestroy(e->aux2);
    (void)hipStreamDestroy(e->stream);
    delete e;
}

const char *bh_engine_last_error(const bh_engine *e) { return e ? e->err.c_str() : "null engine"; }
void *bh_engine_stream(bh_engine *e) { return e ? (void *)e->stream : nullptr; }

int bh_engine_synchronize(bh_engine *e)
{
    if (!e) return BH_EINVAL;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return BH_OK;
}

int bh_engine_set_instrumentation(bh_engine *e, int timing, int counting)
{
    if (!e) return BH_EINVAL;
    e->timing = timing != 0;
    e->counting = counting != 0;
    return BH_OK;
}

int bh_timing_reset(bh_engine *e)
{
    if (!e) return BH_EINVAL;
    e->ncalls = 0;
    return BH_OK;
}

int bh_timing_collect(bh_engine *e, int *ncalls, double *total_ms, double family_ms[3])
{
    if (!e) return BH_EINVAL;
    double tot = 0.0, fam[3] = {0.0, 0.0, 0.0};
    for (size_t i = 0; i < e->ncalls; ++i) {
        bh_engine::EventSet &s = e->evsets[i];
        HIPCHK(e, hipEventSynchronize(s.ev[7]));
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, s.ev[6], s.ev[7]));
        tot += ms;
        for (int f = 0; f < 3; ++f) {
            if (!s.used[f]) continue;
            float fm = 0.f;
            HIPCHK(e, hipEventElapsedTime(&fm, s.ev[2 * f], s.ev[2 * f + 1]));
            fam[f] += fm;
        }
    }
    if (ncalls) *ncalls = (int)e->ncalls;
    if (total_ms) *total_ms = tot;
    if (family_ms)
        for (int f = 0; f < 3; ++f) family_ms[f] = fam[f];
    return BH_OK;
}

int bh_timing_steps(bh_engine *e, int max, double *step_ms, int *n)
{
    if (!e || !step_ms || !n || max < 0) return BH_EINVAL;
    int k = 0;
    for (size_t i = 0; i < e->ncalls && k < max; ++i, ++k) {
        bh_engine::EventSet &s = e->evsets[i];
        HIPCHK(e, hipEventSynchronize(s.ev[7]));
        float ms = 0.f;
        if (i + 1 < e->ncalls) HIPCHK(e, hipEventElapsedTime(&ms, s.ev[6], e->evsets[i + 1].ev[6]));
        else HIPCHK(e, hipEventElapsedTime(&ms, s.ev[6], s.ev[7]));
        step_ms[k] = ms;
    }
    *n = k;
    return BH_OK;
}

int bh_debug_counters(bh_engine *e, uint64_t out[16])
{
    if (!e || !out) return BH_EINVAL;
    if (!e->counter.p) return fail(e, BH_EINVAL, "counting was never enabled");
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(out, e->counter.p, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return BH_OK;
}

int bh_debug_trace(bh_engine *e, uint64_t *out, int nwaves)
{
    if (!e || !out || nwaves < 0 || nwaves > BH_TRACE_WAVES) return BH_EINVAL;
    if (!e->counter.p) return fail(e, BH_EINVAL, "counting was never enabled");
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(out, (uint64_t *)e->counter.p + BH_COUNTER_WORDS, (size_t)4 * nwaves * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return BH_OK;
}

int bh_last_neval(bh_engine *e, uint64_t *neval)
{
    if (!e || !neval) return BH_EINVAL;
    if (e->neval_pending) {
        unsigned long long v = 0;
        HIPCHK(e, hipMemcpy(&v, e->counter.p, sizeof(v), hipMemcpyDeviceToHost));
        e->last_neval = v;
        e->neval_pending = false;
    }
    *neval = e->last_neval;
    return BH_OK;
}

int bh_swd_batch(bh_engine *e, int memspace, void *stream, int B, int Lmax, const int32_t *nlay,
                 const double *h, const double *vp, const double *vs, const double *rho,
                 ptrdiff_t sl, ptrdiff_t sb, int K, const double *periods, int iwave, int igr,
                 int mode, int flsph, double *vel, int32_t *err)
{
    int rc;
    if ((rc = check_models(e, B, Lmax, sl, sb))) return rc;
    if ((rc = swd_supported(e, K, iwave, mode, flsph))) return rc;
    if (!nlay || !h || !vp || !vs || !rho || !periods || !vel || !err) return fail(e, BH_EINVAL, "null argument");
    if (B == 0) return BH_OK;
    HIPCHK(e, hipSetDevice(e->device));
    if (memspace == BH_DEVICE) {
        hipStream_t st = stream ? (hipStream_t)stream : e->stream;
        Staged m{nlay, h, vp, vs, rho, nullptr, nullptr};
        call_begin(e, st);
        SwdJob job{K, iwave, igr, K, periods, vel, err, mode, flsph};
        rc = launch_swd_jobs(e, st, B, Lmax, m, sl, sb, 1, &job);
        call_end(e, st);
        return rc;
    }
    hipStream_t st = e->stream;
    Staged m{};
    if ((rc = stage_models(e, B, Lmax, sl, sb, nlay, h, vp, vs, rho, nullptr, nullptr, m))) return rc;
    if ((rc = ensure(e, e->periods, (size_t)BH_MAX_PERIODS * sizeof(double)))) return rc;
    if ((rc = ensure(e, e->vel, (size_t)B * K * sizeof(double)))) return rc;
    if ((rc = ensure(e, e->errb, (size_t)B * sizeof(int32_t)))) return rc;
    HIPCHK(e, hipMemcpyAsync(e->periods.p, periods, (size_t)K * sizeof(double), hipMemcpyHostToDevice, st));
    call_begin(e, st);
    SwdJob job{K, iwave, igr, K, (const double *)e->periods.p, (double *)e->vel.p, (int32_t *)e->errb.p, mode, flsph};
    rc = launch_swd_jobs(e, st, B, Lmax, m, sl, sb, 1, &job);
    call_end(e, st);
    if (rc) return rc;
    HIPCHK(e, hipMemcpyAsync(vel, e->vel.p, (size_t)B * K * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipMemcpyAsync(err, e->errb.p, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    return BH_OK;
}

int bh_rf_batch(bh_engine *e, int memspace, void *stream, int B, int Lmax, const int32_t *nlay,
                const double *h, const double *vp, const double *vs, const double *rho,
                const double *qp, const double *qs, ptrdiff_t sl, ptrdiff_t sb, double p,
                double gauss, int nsamp, double fsamp, double tshift, double nsv, int waveno,
                int nkeep, double *rf)
{
    int rc;
    if ((rc = check_models(e, B, Lmax, sl, sb))) return rc;
    if ((rc = rf_args_ok(e, nsamp, nkeep, gauss, fsamp, waveno))) return rc;
    if (!nlay || !h || !vp || !vs || !rho || !rf) return fail(e, BH_EINVAL, "null argument");
    if (B == 0 || nkeep == 0) return BH_OK;
    HIPCHK(e, hipSetDevice(e->device));
    if (memspace == BH_DEVICE) {
        hipStream_t st = stream ? (hipStream_t)stream : e->stream;
        Staged m{nlay, h, vp, vs, rho, qp, qs};
        call_begin(e, st);
        rc = launch_rf(e, st, B, Lmax, m, sl, sb, p, gauss, nsamp, fsamp, tshift, nsv, waveno, nkeep, rf, nkeep);
        call_end(e, st);
        return rc;
    }
    hipStream_t st = e->stream;
    Staged m{};
    if ((rc = stage_models(e, B, Lmax, sl, sb, nlay, h, vp, vs, rho, qp, qs, m))) return rc;
    if ((rc = ensure(e, e->rf, (size_t)B * nkeep * sizeof(double)))) return rc;
    call_begin(e, st);
    rc = launch_rf(e, st, B, Lmax, m, sl, sb, p, gauss, nsamp, fsamp, tshift, nsv, waveno, nkeep,
                   (double *)e->rf.p, nkeep);
    call_end(e, st);
    if (rc) return rc;
    HIPCHK(e, hipMemcpyAsync(rf, e->rf.p, (size_t)B * nkeep * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    return BH_OK;
}

int bh_targets_set(bh_engine *e, int nt, const bh_target_desc *td)
{
    if (!e) return BH_EINVAL;
    if (nt < 0 || nt > BH_MAX_TARGETS || (nt > 0 && !td)) return fail(e, BH_EINVAL, "nt must be 0..8");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (auto &t : e->targets) {
        release_target(t);
    }
    e->targets.clear();
    e->nt = 0;
    e->ldy = 0;
    int off = 0;
    std::vector<TargetHost> tmp((size_t)nt);
    for (int i = 0; i < nt; ++i) {
        const bh_target_desc &d = td[i];
        TargetHost &t = tmp[(size_t)i];
        t.d = d;
        t.off = off;
        int rc = BH_OK;
        if (d.n < 1 || !d.yobs) rc = fail(e, BH_EINVAL, "target needs n >= 1 and yobs");
        if (!rc && d.law != BH_LAW_NOCORR && d.law != BH_LAW_NOCORR_SCALED && d.law != BH_LAW_EXP && d.law != BH_LAW_GAUSS)
            rc = fail(e, BH_EINVAL, "unknown covariance law");
        if (!rc && d.kind == BH_TARGET_SWD) {
            if (!d.x) rc = fail(e, BH_EINVAL, "SWD target needs periods x");
            if (!rc) rc = swd_supported(e, d.n > BH_MAX_PERIODS ? BH_MAX_PERIODS : d.n, d.iwave, d.mode, d.flsph);
        } else if (!rc && d.kind == BH_TARGET_RF) {
            rc = rf_args_ok(e, d.nsamp, d.n, d.gauss, d.fsamp, d.waveno);
        } else if (!rc && d.kind == BH_TARGET_USER) {
            /* likelihood-only target */
        } else if (!rc) {
            rc = fail(e, BH_EINVAL, "unknown target kind");
        }
        if (!rc && d.law == BH_LAW_NOCORR_SCALED && !d.yerr) rc = fail(e, BH_EINVAL, "scaled-error law needs yerr");
        if (!rc && d.law == BH_LAW_GAUSS && !d.rinv) rc = fail(e, BH_EINVAL, "Gauss law needs rinv");
        const size_t nb = (size_t)d.n * sizeof(double);
        if (!rc) rc = ensure(e, t.yobs, nb);
        if (!rc && hipMemcpy(t.yobs.p, d.yobs, nb, hipMemcpyHostToDevice) != hipSuccess) rc = fail(e, BH_EHIP, "copy yobs");
        if (!rc && d.kind == BH_TARGET_SWD) {
            rc = ensure(e, t.x, nb);
            if (!rc && hipMemcpy(t.x.p, d.x, nb, hipMemcpyHostToDevice) != hipSuccess) rc = fail(e, BH_EHIP, "copy x");
            t.kfwd = d.n;
            if (!rc && d.n > BH_MAX_PERIODS) {
                // surf96_modsw.py:35-43: np.linspace(min, max, 60), velocities interpolated back (:119-122)
                double lo = d.x[0], hi = d.x[0];
                for (int k = 1; k < d.n; ++k) {
                    lo = d.x[k] < lo ? d.x[k] : lo;
                    hi = d.x[k] > hi ? d.x[k] : hi;
                }
                double g[BH_MAX_PERIODS];
                const double step = (hi - lo) / (double)(BH_MAX_PERIODS - 1);
                for (int k = 0; k < BH_MAX_PERIODS; ++k) g[k] = (double)k * step + lo; // numpy.linspace
                g[BH_MAX_PERIODS - 1] = hi;
                t.kfwd = BH_MAX_PERIODS;
                rc = ensure(e, t.x60, sizeof(g));
                if (!rc && hipMemcpy(t.x60.p, g, sizeof(g), hipMemcpyHostToDevice) != hipSuccess) rc = fail(e, BH_EHIP, "copy x60");
            }
        }
        if (!rc && d.law == BH_LAW_NOCORR_SCALED) { // Targets.py:124-128
            std::vector<double> se(d.yerr, d.yerr + d.n);
            double mn = se[0];
            for (double v : se) mn = v < mn ? v : mn;
            double prod = 1.0;
            for (double &v : se) {
                v = v / mn;
                prod *= v;
            }
            t.logdet_extra = std::log(prod);
            rc = ensure(e, t.yerr_scaled, nb);
            if (!rc && hipMemcpy(t.yerr_scaled.p, se.data(), nb, hipMemcpyHostToDevice) != hipSuccess) rc = fail(e, BH_EHIP, "copy yerr");
        }
        if (!rc && d.law == BH_LAW_GAUSS) {
            t.logdet_extra = d.logdet_r;
            rc = ensure(e, t.rinv, nb * (size_t)d.n);
            if (!rc && hipMemcpy(t.rinv.p, d.rinv, nb * (size_t)d.n, hipMemcpyHostToDevice) != hipSuccess) rc = fail(e, BH_EHIP, "copy rinv");
        }
        if (rc) {
            for (auto &u : tmp) release_target(u);
            return rc;
        }
        // the engine keeps no host pointers
        t.d.x = t.d.yobs = t.d.yerr = t.d.rinv = nullptr;
        off += d.n;
    }
    e->targets.swap(tmp);
    e->nt = nt;
    e->ldy = off;
    e->err_t_nt = e->err_t_B = -1;
    return BH_OK;
}

int bh_evaluate_batch(bh_engine *e, int memspace, void *stream, int B, int Lmax,
                      const int32_t *nlay, const double *h, const double *vp, const double *vs,
                      const double *rho, ptrdiff_t sl, ptrdiff_t sb, const double *noise,
                      double *logL, double *misfits, int32_t *err, double *ymod)
{
    int rc;
    if ((rc = check_models(e, B, Lmax, sl, sb))) return rc;
    if (e->nt < 1) return fail(e, BH_EINVAL, "no targets registered (bh_targets_set)");
    for (const auto &T : e->targets)
        if (T.d.kind == BH_TARGET_USER) return fail(e, BH_EINVAL, "a BH_TARGET_USER target has no forward model: use bh_loglike_batch");
    if (!nlay || !h || !vp || !vs || !noise || !logL || !misfits || !err) return fail(e, BH_EINVAL, "null argument");
    if (B == 0) return BH_OK;
    HIPCHK(e, hipSetDevice(e->device));
    const int nt = e->nt, ldy = e->ldy;
    const bool host = (memspace != BH_DEVICE);
    hipStream_t st = (!host && stream) ? (hipStream_t)stream : e->stream;
    Staged m{nlay, h, vp, vs, rho, nullptr, nullptr};
    const double *noise_d = noise;
    double *logL_d = logL, *misf_d = misfits, *ymod_d = ymod;
    int32_t *err_d = err;
    if (host) {
        if ((rc = stage_models(e, B, Lmax, sl, sb, nlay, h, vp, vs, rho, nullptr, nullptr, m))) return rc;
        if ((rc = ensure(e, e->noise, (size_t)B * 2 * nt * sizeof(double)))) return rc;
        if ((rc = ensure(e, e->logL, (size_t)B * sizeof(double)))) return rc;
        if ((rc = ensure(e, e->misfits, (size_t)B * (nt + 1) * sizeof(double)))) return rc;
        if ((rc = ensure(e, e->errb, (size_t)B * sizeof(int32_t)))) return rc;
        HIPCHK(e, hipMemcpyAsync(e->noise.p, noise, (size_t)B * 2 * nt * sizeof(double), hipMemcpyHostToDevice, st));
        noise_d = (const double *)e->noise.p;
        logL_d = (double *)e->logL.p;
        misf_d = (double *)e->misfits.p;
        err_d = (int32_t *)e->errb.p;
        ymod_d = nullptr;
    }
    if (!ymod_d) {
        if ((rc = ensure(e, e->ymod, (size_t)B * ldy * sizeof(double)))) return rc;
        ymod_d = (double *)e->ymod.p;
    }
    {
        const size_t cap0 = e->err_t.cap;
        if ((rc = ensure(e, e->err_t, (size_t)nt * B * sizeof(int32_t)))) return rc;
        if (e->err_t.cap != cap0) e->err_t_nt = e->err_t_B = -1; // a new buffer: not zeroed yet
    }
    if (!m.rho) { // rho = 0.32 vp + 0.77 (Targets.py:319)
        const size_t nel = span_elems(B, Lmax, sl, sb);
        if ((rc = ensure(e, e->rho, nel * sizeof(double)))) return rc;
        hipLaunchKernelGGL(rho_from_vp_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, nel, m.vp, (double *)e->rho.p);
        m.rho = (const double *)e->rho.p;
    }
    call_begin(e, st);
    // per-target failure flags [nt][B]: the dispersion kernels write every entry of their target's row on every call,
    // nothing writes the rows of the other targets -- they are zeroed once per (nt, B) layout, not once per call
    const bool always_zero = bh_tuning().err_memset != 0;
    if (always_zero || e->err_t_nt != nt || e->err_t_B != B) {
        HIPCHK(e, hipMemsetAsync(e->err_t.p, 0, (size_t)nt * B * sizeof(int32_t), st));
        e->err_t_nt = nt;
        e->err_t_B = B;
    }
    LikeKernelArgs la{};
    la.B = B; la.nt = nt; la.ldy = ldy; la.ymod = ymod_d; la.err_t = (const int32_t *)e->err_t.p;
    la.noise = noise_d; la.logL = logL_d; la.misfits = misf_d; la.err = err_d;
    SwdJob jobs[BH_MAX_TARGETS];
    int njobs = 0;
    for (int t = 0; t < nt; ++t) {
        TargetHost &T = e->targets[(size_t)t];
        const bh_target_desc &d = T.d;
        if (d.kind == BH_TARGET_SWD) {
            int32_t *errp = (int32_t *)e->err_t.p + (size_t)t * B;
            if (T.kfwd == d.n) {
                jobs[njobs++] = SwdJob{d.n, d.iwave, d.igr, ldy, (const double *)T.x.p, ymod_d + T.off, errp, d.mode, d.flsph};
            } else { // > 60 periods: run on the 60-point grid, interpolate afterwards
                if ((rc = ensure(e, T.vel60, (size_t)B * T.kfwd * sizeof(double)))) return rc;
                jobs[njobs++] = SwdJob{T.kfwd, d.iwave, d.igr, T.kfwd, (const double *)T.x60.p, (double *)T.vel60.p, errp, d.mode, d.flsph};
            }
        }
    }
    // The receiver-function kernels are independent of the dispersion kernel and write other columns
    // of ymod: they run on a second stream so that their (throughput-bound) wavefronts fill the issue
    // slots the (latency-bound) dispersion wavefronts leave idle.  fork -> [swd | rf] -> join -> like.
    bool have_rf = false;
    for (int t = 0; t < nt; ++t) have_rf = have_rf || e->targets[(size_t)t].d.kind == BH_TARGET_RF;
    const bool fork = have_rf && njobs > 0 && e->overlap_rf;
    hipStream_t rst = fork ? e->aux : st;
    if (fork) {
        HIPCHK(e, hipEventRecord(e->ev_fork, st));
        HIPCHK(e, hipStreamWaitEvent(e->aux, e->ev_fork, 0));
    }
    // Start gate.  The dispersion group kernel counts on ALL its wavefronts being co-resident (one round of wavefronts,
    // two per SIMD); an RF workgroup that takes a wave slot first displaces a dispersion wavefront for a whole lifetime
    // (3.6 -> 7 ms when the 17.7 KB workgroups of round 3 slipped into the LDS the dispersion wavefronts leave free), and
    // even the small coefficient kernel, dispatched while the dispersion kernel's workgroups are being placed, stretched
    // that kernel's span by 0.3 ms in round 2.  The kernel's workgroups therefore count themselves in `started` as they
    // begin, and the RF stream waits for the count (hipStreamWaitValue32) before anything of the RF is dispatched:
    // c3 4.15 -> 4.04 ms, the dispersion kernel's time inside c3 = its time in c2 (3.62 ms).
    // (Round 6: the dispersion kernel of the default settings allocates 200-208 registers and the synthesis kernel 88, so an RF
    // wavefront becomes resident beside the two dispersion wavefronts of a SIMD and takes the issue slots they leave idle; the
    // 96-register "beside" build of rounds 3-5 -- docs/HISTORY.md -- is gone.)
    const bool want_gate = fork && e->started != nullptr;
    if (e->started != nullptr && e->started_expected > 0x70000000u) { // (the counter is cumulative: rewind it long before it wraps)
        HIPCHK(e, hipStreamSynchronize(st));
        HIPCHK(e, hipStreamSynchronize(e->aux));
        HIPCHK(e, hipMemset(e->started, 0, sizeof(unsigned)));
        e->started_expected = 0;
    }
    // (RF wavefronts move in beside the last dispersion wavefronts of a SIMD as its short ones end: the dispersion
    // wavefronts' unfavoured phase runs at priority 1 then, above the RF's 0)
    e->swd_prio_low_now = want_gate ? e->swd_prio_low : 0;
    e->last_swd = SwdLaunchInfo{};
    rc = launch_swd_jobs(e, st, B, Lmax, m, sl, sb, njobs, jobs);
    e->swd_prio_low_now = 0;
    if (rc) return rc;
    e->rf_gated_now = false;
    if (want_gate && e->last_swd.workgroups > 0) {
        e->rf_gated_now = bh_tuning().rf_keep_floor == 0;
        // all workgroups of the launch, or -- a launch of more workgroups than the chip holds at once (2048 wavefronts) --
        // as many as can be resident together (the rest start as others end: waiting for them would be waiting for the kernel)
        const unsigned resident = 2048u / (unsigned)(e->last_swd.lds > 0 && e->last_swd.workgroups > 0 ? (e->last_swd_wpb > 0 ? e->last_swd_wpb : 2) : 2);
        const unsigned need = e->last_swd.workgroups < resident ? e->last_swd.workgroups : resident;
        HIPCHK(e, hipStreamWaitValue32(e->aux, e->started, e->started_expected - e->last_swd.workgroups + need, hipStreamWaitValueGte,
                                       0xffffffffu));
    }
    for (int t = 0; t < nt; ++t) {
        TargetHost &T = e->targets[(size_t)t];
        const bh_target_desc &d = T.d;
        if (d.kind == BH_TARGET_SWD && T.kfwd != d.n)
            bh_launch_interp(B, T.kfwd, (const double *)T.x60.p, (const double *)T.vel60.p, T.kfwd, d.n,
                             (const double *)T.x.p, ymod_d + T.off, ldy, st);
        T.fused = false;
        if (d.kind != BH_TARGET_RF) continue;
        // Fused likelihood ("write nothing if the likelihood is fused"): the caller did not ask for the
        // synthetics and the target's law needs only sums over the trace -- the synthesis kernel forms them from the samples
        // in LDS (like_kernel's order: the same bits) and writes four numbers per model instead of the trace.
        T.fused = !ymod && bh_tuning().rf_no_fuse == 0 && (d.law == BH_LAW_NOCORR || d.law == BH_LAW_EXP) && bh_tuning().rf_threads != 128;
        if (T.fused && (rc = ensure(e, T.sums, (size_t)B * 4 * sizeof(double)))) return rc;
        rc = launch_rf(e, rst, B, Lmax, m, sl, sb, d.p_s_per_deg, d.gauss, d.nsamp, d.fsamp, d.tshift,
                       d.nsv, d.waveno, d.n, ymod_d + T.off, ldy, fork, T.fused ? (const double *)T.yobs.p : nullptr,
                       T.fused ? (double *)T.sums.p : nullptr);
        if (rc) return rc;
    }
    if (fork) {
        HIPCHK(e, hipEventRecord(e->ev_join, e->aux));
        HIPCHK(e, hipStreamWaitEvent(st, e->ev_join, 0));
    }
    for (int t = 0; t < nt; ++t)
        if ((rc = prepare_like_target(e, st, B, ldy, ymod_d, e->targets[(size_t)t], la.t[t]))) return rc;
    ev_begin(e, 2, st);
    bh_launch_like(la, st);
    ev_end(e, 2, st);
    call_end(e, st);
    HIPCHK(e, hipGetLastError());
    if (host) {
        HIPCHK(e, hipMemcpyAsync(logL, logL_d, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipMemcpyAsync(misfits, misf_d, (size_t)B * (nt + 1) * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipMemcpyAsync(err, err_d, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (ymod) HIPCHK(e, hipMemcpyAsync(ymod, ymod_d, (size_t)B * ldy * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipStreamSynchronize(st));
    }
    return BH_OK;
}

int bh_loglike_batch(bh_engine *e, int memspace, void *stream, int B, const double *ymod,
                     const int32_t *failflags, const double *noise, double *logL, double *misfits,
                     int32_t *err)
{
    int rc;
    if (!e) return BH_EINVAL;
    if (e->nt < 1) return fail(e, BH_EINVAL, "no targets registered (bh_targets_set)");
    if (B < 0 || !ymod || !noise || !logL || !misfits || !err) return fail(e, BH_EINVAL, "null argument");
    if (B == 0) return BH_OK;
    HIPCHK(e, hipSetDevice(e->device));
    const int nt = e->nt, ldy = e->ldy;
    const bool host = (memspace != BH_DEVICE);
    hipStream_t st = (!host && stream) ? (hipStream_t)stream : e->stream;
    if ((rc = ensure(e, e->err_t, (size_t)nt * B * sizeof(int32_t)))) return rc;
    e->err_t_nt = e->err_t_B = -1; // (this call may fill the flag rows with the caller's: bh_evaluate_batch zeroes them again)
    LikeKernelArgs la{};
    la.B = B; la.nt = nt; la.ldy = ldy;
    la.ymod = ymod; la.noise = noise; la.logL = logL; la.misfits = misfits; la.err = err;
    la.err_t = failflags;
    if (host) {
        if ((rc = ensure(e, e->ymod, (size_t)B * ldy * sizeof(double)))) return rc;
        if ((rc = ensure(e, e->noise, (size_t)B * 2 * nt * sizeof(double)))) return rc;
        if ((rc = ensure(e, e->logL, (size_t)B * sizeof(double)))) return rc;
        if ((rc = ensure(e, e->misfits, (size_t)B * (nt + 1) * sizeof(double)))) return rc;
        if ((rc = ensure(e, e->errb, (size_t)B * sizeof(int32_t)))) return rc;
        HIPCHK(e, hipMemcpyAsync(e->ymod.p, ymod, (size_t)B * ldy * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(e, hipMemcpyAsync(e->noise.p, noise, (size_t)B * 2 * nt * sizeof(double), hipMemcpyHostToDevice, st));
        if (failflags) HIPCHK(e, hipMemcpyAsync(e->err_t.p, failflags, (size_t)nt * B * sizeof(int32_t), hipMemcpyHostToDevice, st));
        la.ymod = (const double *)e->ymod.p; la.noise = (const double *)e->noise.p;
        la.logL = (double *)e->logL.p; la.misfits = (double *)e->misfits.p; la.err = (int32_t *)e->errb.p;
        la.err_t = failflags ? (const int32_t *)e->err_t.p : nullptr;
    }
    if (!la.err_t) {
        HIPCHK(e, hipMemsetAsync(e->err_t.p, 0, (size_t)nt * B * sizeof(int32_t), st));
        la.err_t = (const int32_t *)e->err_t.p;
    }
    call_begin(e, st);
    for (int t = 0; t < nt; ++t) {
        e->targets[(size_t)t].fused = false; // (the caller's synthetics, not sums of an earlier fused call)
        if ((rc = prepare_like_target(e, st, B, ldy, la.ymod, e->targets[(size_t)t], la.t[t]))) return rc;
    }
    ev_begin(e, 2, st);
    bh_launch_like(la, st);
    ev_end(e, 2, st);
    call_end(e, st);
    HIPCHK(e, hipGetLastError());
    if (host) {
        HIPCHK(e, hipMemcpyAsync(logL, la.logL, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipMemcpyAsync(misfits, la.misfits, (size_t)B * (nt + 1) * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipMemcpyAsync(err, la.err, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipStreamSynchronize(st));
    }
    return BH_OK;
}

int bh_probe_math(bh_engine *e, int op, int n, const double *in, double *out)
{
    if (!e || n < 0 || !in || !out) return BH_EINVAL;
    if (n == 0) return BH_OK;
    int rc;
    HIPCHK(e, hipSetDevice(e->device));
    const size_t nin = (op == 6 || op == 7) ? (size_t)2 * n : (size_t)n; // division probes read pairs
    if ((rc = ensure(e, e->probe_in, nin * sizeof(double)))) return rc;
    if ((rc = ensure(e, e->probe_out, (size_t)n * sizeof(double)))) return rc;
    HIPCHK(e, hipMemcpyAsync(e->probe_in.p, in, nin * sizeof(double), hipMemcpyHostToDevice, e->stream));
    bh_launch_probe(op, n, (const double *)e->probe_in.p, (double *)e->probe_out.p, e->stream);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipMemcpyAsync(out, e->probe_out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return BH_OK;
}

} // extern "C"
