// bayhunter_amd/csrc/bh_device.h -- shared declarations of the gfx950 kernels.
//
// Everything in csrc/ is written for CDNA4 (MI355X, wave64) only and is compiled with
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// -ffp-contract=off matters: the reference's Fortran/C++ round every product and sum
// separately (x86-64 baseline, no FMA); the root search of surf96 branches on signs and on
// 1e-6-relative comparisons of those values, so the kernels keep the same rounding points.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define BH_WAVE 64
// debug counter block of the dispersion kernels: BH_COUNTER_WORDS counters + 4 words per traced wavefront
#define BH_COUNTER_WORDS 16
#define BH_TRACE_WAVES 16384
#define BH_DEBUG_WORDS (BH_COUNTER_WORDS + 4 * BH_TRACE_WAVES)
#define BH_BOARD_WORDS (2 * 32768) // progress board: 2 wave slots x (8 XCCs x 4096 SIMD keys)

// ---- IEEE division with the denominator-only work factored out ----------------------------------
// hipcc expands the f64 `a / b` to   d = div_scale(b), n = div_scale(a), r = rcp(d),
// two Newton steps on r, q = n*r, rem = fma(-d, q, n), div_fmas(rem, r, q), div_fixup.
// When neither operand needs the power-of-two pre-scaling (both magnitudes in
// [2^-400, 2^400] here, far inside the hardware's trigger points) d = b, n = a, div_fmas is a
// plain fma and div_fixup returns its input, so  bh_quot(a, b, bh_rcp_refined(b))  is the SAME
// instruction sequence and returns the same bits as a / b -- but the five denominator-only
// instructions can be shared by several divisions or hoisted out of a dependent chain.
// tests/test_gpu_swd.py::test_shared_reciprocal_division_is_exact checks this on the device.
__device__ __forceinline__ double bh_rcp_refined(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
__device__ __forceinline__ double bh_quot(double a, double b, double r)
{
    const double q = a * r;
    const double rem = __builtin_fma(-b, q, a);
    return __builtin_fma(rem, r, q);
}
// |x| in [2^-400, 2^400]  (biased exponent 623..1423); false for 0, denormals, inf, NaN
__device__ __forceinline__ bool bh_div_safe(double x)
{
    const unsigned ex = ((unsigned)__double2hiint(x) >> 20) & 0x7ffu;
    return (ex - 623u) <= 800u;
}

// hipFuncAttributeMaxDynamicSharedMemorySize applies per DEVICE: remember per device (bit d of *done) that a set of kernels
// has been allowed `bytes` of dynamic LDS.  (A process-wide flag left the second GPU of a process without it: ADVICE r03.)
#include <atomic>
inline bool bh_allow_big_lds(std::atomic<unsigned long long> *done, const void *const *kernels, int nkernels, int bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return false;
    const unsigned long long bit = 1ull << dev;
    if (done->load(std::memory_order_acquire) & bit) return true;
    for (int k = 0; k < nkernels; ++k)
        if (hipFuncSetAttribute(kernels[k], hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done->fetch_or(bit, std::memory_order_release);
    return true;
}

struct SwdKernelArgs {
    int B, Lmax, K, igr, mode;
    const int32_t *nlay;
    const int32_t *perm; // optional: models in the order the wavefronts take them (deepest first), see bh_launch_order
    const double *h, *vp, *vs, *rho;
    ptrdiff_t sl, sb; // element strides: layer, model
    const double *periods;
    double *vel;      // [B][ldv] (+ column offset already applied)
    int ldv;          // row stride of vel in elements
    int32_t *err;     // [B]
    unsigned long long *neval; // optional global counter of secular evaluations (may be null)
    int look;         // trial velocities per round and model = lanes per model (1, 2, 4, 8 or 16), see SearchT::candidate
    int fair;         // alternate the issue priority of the two wavefronts of a SIMD (scheduling only)
    double *nev_high; // work array [bh_swd_nev_high_doubles(B, look)]: Neville orders the kernel does not keep in LDS
    int fast;         // 1: the build with the short refinement (SearchT<.., FAST>; phase-velocity targets take it)
    int counted;      // 1: Love scans skip the steps a mode count proves empty (SearchT: the counted scan; same bits)
    int farith;       // 1: a launch of the short refinement (fundamental-mode phase velocities) computes with the fast arithmetic (swd_fa.h)
    int32_t *gcount, *glist; // short refinement: models its guard fired on are appended here (count, indices), see SearchT
    // A group velocity's two chains of roots (SearchT: F_CHAIN_A / F_ONE).  igr = 2: the chain of first roots, unrounded into vel,
    // the value getsol keeps from the mode's first search into first[model].  second = 1 (igr = 1): B = Bm x K searches, entry
    // v = the second root of period v / Bm of model v % Bm, which reads vel and first and writes the group velocity (err untouched).
    int second, Bm;
    double *first;
};

void bh_launch_swd(const SwdKernelArgs &a, int iwave, hipStream_t stream);
// perm[0..B) = model indices sorted by layer count, deepest first: wavefronts then hold models of (nearly)
// one depth -- no masked layers, and the long-running deep models start first.  Results do not depend on it.
void bh_launch_order(int B, const int32_t *nlay, int32_t *perm, int Lcut, int32_t *split, hipStream_t stream);
size_t bh_swd_lds_bytes(int Lmax, int K, int mode);
size_t bh_swd_nev_high_doubles(int B, int look);

// group kernel: G lanes per model, all dispersion targets of a call in one launch
struct SwdTarget {
    int iwave, igr, K, ldv, mode;
    int look; // trial velocities per round for this target's wavefronts (>= 1), see SearchT::candidate
    int inlook; // Love only: further trials inside a lane group (1..4), see swd_group_kernel
    const double *h, *vp, *vs, *rho; // model arrays this target reads (earth-flattened copies when flsph = 1)
    ptrdiff_t sl, sb;
    const double *periods;
    double *vel;  // [B][ldv] (+ column offset already applied)
    int32_t *err; // [B]
    const int32_t *perm; // optional: this target's own processing order (bh_launch_pair_order); else SwdMultiArgs::perm
    const int32_t *count; // optional (device): the launch covers the first *count entries of the processing order only (the
                          // re-run of the models the short refinement's guard fired on: perm = that list)
    int32_t *gcount, *glist; // short refinement: models its guard fired on are appended here (count, indices), see SearchT
    int refseq;              // 1: in a launch with the short refinement this (phase-velocity) target keeps the reference's sequence
    double *first;           // igr = 2 (the chain of a group velocity's first roots, SwdKernelArgs): [B]
};
struct SwdMultiArgs {
    int B, Lmax, ntargets;
    const int32_t *nlay;
    const int32_t *perm; // optional, as in SwdKernelArgs
    // Two classes of models in processing order: [0, split[0]) have more than `Lcut` layers ("deep"),
    // the rest at most Lcut.  Each class gets its own launch with LDS rows for its depth, so that a
    // batch whose array capacity (Lmax) is far above its typical depth still packs many models per
    // wavefront.  Both classes run in ONE launch (blockIdx.z = class) with the same LDS budget per
    // wavefront: the deep class simply takes fewer models per wavefront (more lanes per model).
    // split == nullptr: one class (index 1), rows = Lmax.  rows[] / lanes[] are set by the launcher.
    const int32_t *split;
    int Lcut;
    int rows[2], lanes[2]; // per class (0 = deep, 1 = the rest): LDS rows per model, lanes per model (G)
    int wg_n0, wg_n1;      // set by the launcher: > 0 = one-dimensional grid of two targets, interleaved (wavefront counts)
    unsigned long long *neval;
    unsigned *board;  // optional: progress board of the group kernel, 2 words per physical SIMD (BH_BOARD_WORDS), see the kernel
    unsigned stamp;   // launch stamp (16 bits) that marks this launch's entries of the board
    unsigned *started; // optional: every workgroup adds 1 when it starts (cumulative over launches): a second stream waits
                       // for "all workgroups of this launch are resident" before it dispatches work beside them
    int prio_low;      // s_setprio level of a wavefront's unfavoured phase (0; 1 when receiver-function wavefronts at 0 run beside it)
    int fast;          // 1: the build with the short refinement (SearchT<.., FAST>; phase-velocity targets take it)
    int adapt_ok;      // 1: a launch of one model per wavefront may let every wavefront size its lane groups and trials for its
                       // own model (swd_group_kernel<.., ADAPT>); 0: the caller fixed lanes or trials (experiments)
    int counted;       // 1: Love scans skip the steps a mode count proves empty (SearchT: the counted scan; same bits)
    int rerun;         // 1: the launch re-runs listed models (SwdTarget::count): plain two-dimensional grid, no SIMD pairing
    int farith;        // 1: launches in which every target takes the short refinement evaluate the secular functions with the fast
                       //    arithmetic (swd_fa.h, swd_group_kernel<.., FA>); set to what took effect by the launcher
    int restart;       // 1: in a launch of one model per wavefront a model the guard fires on starts again with the reference's
                       //    sequence in its own wavefront (the build with both sequences) instead of being listed for a re-run launch
    SwdTarget t[8];
};
int bh_swd_pick_group(int B, int ntargets, int Lmax);
double bh_swd_plan(int B, int Lmax, int ntargets, const int *iwave, int Gforce, int *G, int *look);
size_t bh_swd_group_lds_bytes(int G, int J, int Lmax, int Kmax, int maxmode);
// Work space of the SIMD-pairing order (swd_kernel.hip: pair_order_kernel; swd_group_kernel.hip: the launcher).  The
// dispersion kernel's time is that of its slowest SIMD, a SIMD's time follows the SUM of the root-search lengths of the
// two wavefronts it holds, and which wavefronts of a launch share a SIMD is a fixed function of their grid index; so the
// models are dealt to the wavefronts by PREDICTED search length such that every SIMD gets a long and a short wavefront
// and the SIMDs that hold one wavefront only get the longest.  Scheduling only: per-model results do not depend on it.
struct SwdPairWork {
    int ncu = 0;                      // compute units of the device (the placement rule below is per CU)
    int32_t *perm[2] = {nullptr, nullptr};      // device, [B] each: processing order of target 0 / 1
    int32_t *slot_rank[2] = {nullptr, nullptr}; // device, [wavefronts of the target]: rank of the wavefront's load (0 = longest models)
    int cap_perm = 0, cap_rank[2] = {0, 0};
    int key_n0 = -1, key_n1 = -1, key_wpb = -1; // geometry the slot_rank tables were built for
};
struct PairOrderTarget {
    int mpw, nwaves;          // models per wavefront, wavefronts of the target
    const int32_t *slot_rank; // [nwaves - 1]: load rank of every wavefront but the last (which may be partly filled)
    int32_t *perm;            // out [B]
    int xcd;                  // trial-per-lane kernel: > 0 = the order is sorted inside eight blocks of the batch, block x handed to the
                              // wavefronts that run on XCD x (workgroup index mod 8); the value = wavefronts of this target per workgroup (4, or 2
                              // where two targets alternate): every XCD's L2 then sees an eighth of the model arrays
};
bool bh_pair_order_fits(int B);
void bh_launch_pair_order(int B, int Lmax, const int32_t *nlay, const double *vs, ptrdiff_t sl, ptrdiff_t sb, int nt,
                          const PairOrderTarget *tg, hipStream_t stream);
struct SwdLaunchInfo {
    unsigned workgroups; // of the launch (what SwdMultiArgs::started is advanced by)
    long waves;          // wavefronts that do work
    size_t lds;          // bytes per workgroup
    int fast_arith;      // the launch evaluates with the fast arithmetic (SwdMultiArgs::farith took effect)
    int restarts_in_place; // the launch handles guarded models itself (SwdMultiArgs::restart took effect): no re-run launch needed
};
// the launches of the builds with the fast arithmetic (swd_group_fa.hip); called by bh_launch_swd_group
void bh_launch_swd_group_fa(const SwdMultiArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t stream, int redundant, int wave_lds,
                            bool adapt, bool counted, bool cntb);
// the launches of the one-model-per-wavefront builds (swd_group_adapt.hip); called by bh_launch_swd_group
void bh_launch_swd_group_adapt(const SwdMultiArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t stream, int redundant, int wave_lds,
                               int fm, bool pr, bool cn);
// the launch of the build that needs one wavefront per SIMD as its register budget (swd_group_big.hip); called by bh_launch_swd_group
void bh_launch_swd_group_big(const SwdMultiArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t stream, int redundant, int wave_lds);
int bh_launch_swd_group(const SwdMultiArgs &a, int G, hipStream_t stream, SwdLaunchInfo *info = nullptr, int wavefronts_per_workgroup = 2,
                        SwdPairWork *pair = nullptr);
// swd_lean.hip: fundamental-mode phase velocities with the fast arithmetic, one lane per trial velocity (the kernel of the
// engine's default settings for batches up to a few ten thousand models); a.t[t].look = trials per model and round
int bh_swd_lean_trials(int B, int ntargets);
size_t bh_swd_lean_lds_bytes(int J, int Lmax, int Kmax);
int bh_launch_swd_lean(const SwdMultiArgs &a, hipStream_t stream, SwdLaunchInfo *info);
// earth-flattening of a batch (surfdisp96.f:486-553): writes layer-major [Lmax][B] float64 copies
// (binary32-valued) of thickness, vp, vs and the Love / Rayleigh density mappings
void bh_launch_sphere(int B, int Lmax, const int32_t *nlay, const double *h, const double *vp,
                      const double *vs, const double *rho, ptrdiff_t sl, ptrdiff_t sb, double *oh,
                      double *ovp, double *ovs, double *orho_love, double *orho_ray, hipStream_t stream);
// np.interp of B rows from K0 to K1 abscissae (the > 60 periods path of surf96_modsw.py:119-122)
void bh_launch_interp(int B, int K0, const double *x0, const double *y0, int ld0, int K1,
                      const double *x1, double *y1, int ld1, hipStream_t stream);

struct RfKernelArgs {
    int B, Lmax, nsamp, nkeep, waveno;
    const int32_t *nlay;
    const double *h, *vp, *vs, *rho, *qp, *qs; // qp/qs may be null -> 500/225
    ptrdiff_t sl, sb;
    double p_s_per_deg, gauss, fsamp, tshift, nsv;
    double *coef;  // workspace [B][bh_rf_coef_doubles(Lmax)]
    double *rf;    // [B][ldr]
    int ldr;
    // fused likelihood (bh_evaluate_batch without synthetics, nocorr / exponential law): instead of the trace the synthesis
    // kernel writes, per model, the four sums the likelihood needs of it -- sum d^2, sum d_i d_(i+1), d_0, d_(n-1), d = trace -
    // yobs, formed in like_kernel's own order (same bits) -- to sums[B][4]; null: the trace is written
    const double *yobs;
    double *sums;
    double *zwork; // traces beyond BH_RF_MAX_LDS: workspace [B][nsamp / 2] complex (16 bytes each) for the half-length spectra, else null
    int lds_min;   // lower bound of the synthesis kernel's LDS request in bytes (0 = what the trace needs), see bh_engine.hip
    int no_realc;  // experiment switch: 1 = always the general (complex-coefficient) recursion
    int coef_small; // 1: the 96-register build of the coefficient kernel (fused call: resident beside the dispersion wavefronts)
    int no_rot;    // experiment switch: 1 = no rotation of the bins over a workgroup's wavefronts
};
size_t bh_rf_coef_doubles(int Lmax);
// LDS of one workgroup of the synthesis kernel for traces of nsamp samples; a CU has 160 KB, one workgroup may use all
constexpr size_t BH_RF_MAX_LDS = 160 * 1024;
// Longer traces (nsamp > 16384) keep the half-length spectrum in an HBM workspace (RfKernelArgs::zwork) and run the same
// butterflies there; the largest transform served (the second twiddle table, nsamp / 128 entries, stays in LDS)
constexpr int BH_RF_MAX_NSAMP = 1 << 18;
size_t bh_rf_lds_bytes(int nsamp);
int bh_launch_rf(const RfKernelArgs &a, hipStream_t stream); // 0, or -1 when the trace does not fit a workgroup's LDS

struct LikeTargetDev {
    int law, n, off; // off: column offset of this target's samples inside a ymod row
    const double *yobs, *yerr_scaled, *rinv; // device; yerr_scaled = yerr/min(yerr) (law 1)
    double logdet_extra;                     // ln prod(scaled err) (law 1) or ln|R| (law 3)
    const double *quad;                      // law 3: [B][nsplit] column-slab partial sums of d^T R^-1 d
    int nsplit;                              //        (gauss_kernel.hip); null -> in-kernel mat-vec
    const double *pre;                       // laws 0 / 2: [B][4] sums formed by the forward kernel (RfKernelArgs::sums); null -> from ymod
};
int bh_gauss_nsplit(int B, int n);
void bh_launch_gauss_quad(int B, int n, int ldy, const double *ymod, const double *yobs,
                          const double *rinv, int nsplit, double *partial, hipStream_t stream);
struct LikeKernelArgs {
    int B, nt, ldy;
    const double *ymod; // [B][ldy]
    const int32_t *err_t; // [nt][B] per-target forward-model failure flags
    const double *noise; // [B][2*nt]
    LikeTargetDev t[8];
    double *logL;    // [B]
    double *misfits; // [B][nt+1]
    int32_t *err;    // [B]
};
void bh_launch_like(const LikeKernelArgs &a, hipStream_t stream);

void bh_launch_probe(int op, int n, const double *in, double *out, hipStream_t stream);

// out = [val n][bound n][certified n]
