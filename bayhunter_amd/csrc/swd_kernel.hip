// bayhunter_amd/csrc/swd_kernel.hip -- Rayleigh/Love phase & group dispersion on gfx950.
//
// Replaces the reference's surfdisp96 (src/extensions/surfdisp96.f:55-360 and the routines it
// calls) for a batch of models.  No MFMA: the work is a scalar FP64 recurrence (5-vector x 5x5
// compound matrix per layer for Rayleigh, 2-vector for Love) inside a data-dependent root search.
// Two kernels return identical bits:
//   swd_kernel<1|2>     one wavefront lane = one model (batches that fill the chip by themselves);
//   swd_group_kernel    G lanes = one model, J such groups evaluating the trial velocities the search
//                       will most probably ask for next (everything smaller) -- see the block comment
//                       above it; bh_swd_plan picks the mapping and the look-ahead per launch.
//
// Design points
//   * EVALUATION-SYNCHRONOUS STATE MACHINE.  The reference's control flow is
//       for period: bracket-step until sign change; refine (bisection / inverse Neville)
//     and the number of secular-function evaluations differs per model and per period.  A
//     literal SIMT translation would make every lane wait for the slowest lane in every inner
//     loop.  Here each model keeps an explicit search state (period index, which root, bracket,
//     Neville table, continuation tag) and the wavefront's loop body is exactly ONE secular
//     evaluation for all its models followed by a state transition.  Models drift apart in
//     period index freely; the wave ends when its slowest model has used up its own total, not
//     the sum of per-period maxima.
//   * The model (thickness, vp, vs, rho), rounded to binary32 as the f2py boundary of the
//     reference does, is staged once through LDS from coalesced global loads
//     of the layer-major (vp, vs, rho, h) arrays; the Neville tables x[11], y[11] and the period
//     table live in LDS as well.
//   * Rounding points, the search sequence (start value, 0.005 km/s stepping, direction logic,
//     Neville/bisection decisions, the 1e-6 stop test, which point is returned), the binary32
//     arithmetic of the start value and of the group-velocity formula follow the reference exactly
//     (SURVEY.md App. A), and sin/cos/exp (and log/powf of the flattening transform) are
//     restatements of the host libm the reference links (bh_libm.h): results are bit-identical.
#include "../../include/bh_engine_debug.h"
#include "bh_device.h"
#include "bh_tuning.h"
#include <cmath>
#include <cstdlib>

// glibc-exact exp / sincos (see bh_libm.h): with these the device's secular function is the
// reference's bit for bit -- sqrt and division are correctly rounded on gfx950, and exp / sincos
// were the only operations where the device library (ocml, <= 1 ulp) and the host libm differed.
#define BH_HD __device__ __forceinline__
#define BH_TAB static __device__ const
#include "bh_libm.h"

namespace {
#include "swd_common.h"

constexpr int LANE_TAB_PAD = (LIBM_TAB_BYTES + 15) & ~15;
#ifndef BH_NEV_LO
#define BH_NEV_LO 5 // (a build with 2 sends every interpolation of order >= 2 through the global array: used once to test that path)
#endif
constexpr int NEV_LO = BH_NEV_LO; // Neville orders kept in LDS by kernel 1, see there
// LDS of one wavefront of kernel 1 (without the shared libm tables)
__host__ __device__ inline size_t lane_wave_bytes(int Lmax, int K, int mode)
{
    const size_t b = (size_t)4 * Lmax * 64 * sizeof(float) + (size_t)2 * NEV_LO * 64 * sizeof(double) +
                     (size_t)((K + 1) & ~1) * sizeof(double) + (mode > 1 ? (size_t)2 * K * 64 * sizeof(double) : 0);
    return (b + 15) & ~(size_t)15;
}

// =================================================================================================
// Kernel 1: one lane = one secular evaluation of one model, all layers serial in the lane; everything
// per-layer stays in registers.
//   J = 1   one lane = one model.  Best when the batch alone fills the chip (B*targets/64 >= 2048 waves).
//   J > 1   J neighbouring lanes = one model, lane r evaluating the velocity the search will most probably
//           ask for r requests from now (SearchT::candidate; the look-ahead of the group kernel without its
//           layer-parallel phases): for the batches in between, which give the group kernel several rounds of
//           wavefronts but this kernel, one lane per model, less than one wavefront per SIMD.  Every lane of a
//           model holds the same search state and steps it with the same values: no broadcast.
// Same operations, same bits for every J.
// =================================================================================================
// Residency is bounded by LDS here (a wavefront's model alone is 10 KB at 10 layers), so only the first NEV_LO Neville
// orders are kept in LDS (the higher ones, reached only through runs of consecutive interpolation steps, in a global
// work array) and a workgroup is WPB wavefronts sharing ONE copy of the libm tables: WPB = 1 gives 7 wavefronts per
// CU at 10 layers (10.2 + 5.1 + 0.25 + 5.5 KB; it was 5 with all 11 orders in LDS), WPB = 2 gives 8 -- 5-9 % faster
// from 100 000 models on, 2.5 % slower below (the launcher picks).
// FAST: 0 the reference sequence, 2 the short refinement (a launch is one target); SIMPLE: a fundamental-mode phase-velocity
// launch (SearchT, swd_common.h: no second root, no mode loop)
// FA: the fast arithmetic (swd_fa.h; launches of the short refinement only)
template <int IFUNC, bool LOOK, int LANE_WPB, int FAST, bool SIMPLE, bool FA = false>
__global__ __launch_bounds__(BH_WAVE * LANE_WPB) void swd_kernel(SwdKernelArgs A)
{
    extern __shared__ __align__(16) unsigned char smem_all[];
    const int lane = threadIdx.x & (BH_WAVE - 1);
    const int wave = threadIdx.x / BH_WAVE;
    const int wid = blockIdx.x * LANE_WPB + wave; // this wavefront among all of the launch
    const int J = LOOK ? A.look : 1;      // power of two, 1..16
    const int r = lane & (J - 1);         // this lane's trial
    const int lbase = lane - r;           // first lane of the model
    const int sidx = (wid * BH_WAVE + lane) / J; // position in the processing order
    const bool valid = sidx < A.B;
    const int iv = valid ? (A.perm ? A.perm[sidx] : sidx) : 0;
    // (a launch of second roots, SwdKernelArgs::second: entry iv = period iv / Bm of model iv % Bm -- neighbouring lanes, neighbouring models)
    const bool second = !SIMPLE && FAST == 0 && A.second != 0;
    const int ib = second ? iv % A.Bm : iv;
    const int Lmax = A.Lmax;
    const int K = A.K;

    const LibmTabs LT = stage_libm_tables(smem_all, threadIdx.x, BH_WAVE * LANE_WPB);
    unsigned char *smem = smem_all + LANE_TAB_PAD + (size_t)wave * lane_wave_bytes(Lmax, K, A.mode);
    float *mdl = reinterpret_cast<float *>(smem);                       // [4][Lmax][64]
    double *xs = reinterpret_cast<double *>(smem + (size_t)4 * Lmax * BH_WAVE * sizeof(float));
    double *ys = xs + NEV_LO * BH_WAVE;                                  // [NEV_LO][64] each
    double *per = ys + NEV_LO * BH_WAVE;                                 // [K]
    double *cpl = per + ((K + 1) & ~1);                                  // [2][K][64], only if mode > 1

    for (int k = lane; k < K; k += BH_WAVE) per[k] = A.periods[k];

    // ---- stage the model through LDS, rounding to binary32 like the f2py boundary -----------
    const int mmax = valid ? A.nlay[ib] : 2;
    int mtop = mmax; // wave-wide maximum layer count = loop bound of the secular functions
    for (int off = 32; off > 0; off >>= 1) mtop = max(mtop, __shfl_xor(mtop, off));
    {
        const ptrdiff_t base = (ptrdiff_t)ib * A.sb;
        for (int l = 0; l < Lmax; ++l) {
            float fd = 0.f, fa = 1.f, fb = 1.f, fr = 1.f;
            if (valid && l < mmax) {
                const ptrdiff_t o = base + (ptrdiff_t)l * A.sl;
                fd = (float)A.h[o];
                fa = (float)A.vp[o];
                fb = (float)A.vs[o];
                fr = (float)A.rho[o];
            }
            mdl[(0 * Lmax + l) * BH_WAVE + lane] = fd;
            mdl[(1 * Lmax + l) * BH_WAVE + lane] = fa;
            mdl[(2 * Lmax + l) * BH_WAVE + lane] = fb;
            mdl[(3 * Lmax + l) * BH_WAVE + lane] = fr;
        }
    }
    __syncthreads();
    ModelLds md;
    md.d = mdl + 0 * Lmax * BH_WAVE + lane;
    md.a = mdl + 1 * Lmax * BH_WAVE + lane;
    md.b = mdl + 2 * Lmax * BH_WAVE + lane;
    md.rho = mdl + 3 * Lmax * BH_WAVE + lane;
    const int llw = (md.Bf(0) <= 0.0f) ? 2 : 1;

    SearchT<BH_WAVE, NEV_LO, FAST, SIMPLE> S;
    S.init(md, mmax, valid, A.igr, K, per, xs + lane, ys + lane, A.vel + (size_t)ib * A.ldv, r == 0, A.mode,
           cpl + lane, cpl + (size_t)K * BH_WAVE + lane, IFUNC, A.counted != 0, false, FA);
    {
        const size_t nl = (size_t)gridDim.x * LANE_WPB * BH_WAVE; // lanes of the launch
        double *hx = A.nev_high + (size_t)wid * BH_WAVE + lane;
        S.set_high(hx, hx + (size_t)(NEV_MAX - NEV_LO) * nl, nl);
    }
    if (second) S.template enter_second<IFUNC == 1>(iv / A.Bm, A.first[ib]);

    // Two wavefronts share a SIMD (one of each target when the targets of a call run side by side); at equal
    // priority the hardware serves the OLDER one first and a Rayleigh / Love pair takes as long as the two in
    // sequence (B = 65 536: 25.8 ms ~ 13.4 + 11.5).  Alternating priorities in time slices, the two hardware wave
    // slots in opposite phase, make them overlap: 20.9 ms with slices of 2^18 cycles for such mixed pairs; pairs of
    // the same kind (more than 2048 wavefronts in the call) do best with short slices (2^12).  (Scheduling only.)
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    const unsigned nrounds = hw_id & 1u; // hardware wave slot parity
    const bool fair = A.fair != 0;
    while (__ballot(S.active) != 0ull) {
        if (fair) {
            // time slices of 2^A.fair cycles, the two hardware wave slots in opposite phase (the clock is shared)
            const unsigned slice = (unsigned)(__builtin_readcyclecounter() >> A.fair);
            if (((slice ^ nrounds) & 1u) != 0u) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
        }
        if (!S.active) continue; // (the lanes of a model share its state: they leave together)
        const double omg = S.omega;
        const double cev = LOOK ? S.template candidate<IFUNC == 1>(r) : S.ceval;
        const double wvno = omg / cev;
        double del;
        int nv = -1; // Love: the packed mode count of this evaluation (LoveCount)
        DivRange dr;
        dr.reset();
        if (FA) {
            if (IFUNC == 1) del = fa::love_secular(wvno, omg, md, mmax, llw, mtop, &nv);
            else del = fa::rayleigh_secular(wvno, omg, md, mmax, llw, mtop);
        } else if (IFUNC == 1)
            del = love_secular<false>(wvno, omg, md, mmax, llw, mtop, dr, LT, &nv);
        else
            del = rayleigh_secular<false>(wvno, omg, md, mmax, llw, mtop, dr, LT);
        if (!FA && !dr.ok()) { // operands left the range the fast divisions are exact in: redo verbatim
            if (IFUNC == 1)
                del = love_secular<true>(wvno, omg, md, mmax, llw, mtop, dr, LT, &nv);
            else
                del = rayleigh_secular<true>(wvno, omg, md, mmax, llw, mtop, dr, LT);
        }
        if (!LOOK) {
            S.template advance<IFUNC == 1>(del, nv);
        } else {
            // every lane of the model reads all J (velocity, value) pairs; the search consumes them for as long as
            // its next request is the very velocity (at the same omega) the next lane evaluated
            bool live = true;
            for (int j = 0; j < J; ++j) {
                const double cj = __shfl(cev, lbase + j);
                const double dj = __shfl(del, lbase + j);
                const int nj = (IFUNC == 1) ? __shfl(nv, lbase + j) : -1;
                // trial 0 IS the pending request (consumed unconditionally); a later one only if asked for now
                if (j > 0) live = live && S.active && S.ceval == cj && S.omega == omg;
                if (__ballot(live) == 0ull) break;
                if (live) S.template advance<IFUNC == 1>(dj, nj);
            }
        }
    }
    if (valid && r == 0) {
        if (!second) A.err[ib] = S.errflag;
        if (!SIMPLE && FAST == 0 && A.igr == 2 && A.first != nullptr) A.first[ib] = S.del1st;
        if (FAST != 0 && S.has(S.F_GUARD) && A.gcount != nullptr) { // to be run again with the reference's sequence
            A.glist[atomicAdd(A.gcount, 1)] = ib;
            atomicAdd(A.gcount + 2 * BH_MAX_TARGETS, 1); // (cumulative, for bh_engine_guard_stats)
        }
    }
    if (A.neval != nullptr) {
        // [0] secular evaluations; per wave type ([8] Rayleigh / [9] Love) evaluations and ([10] / [11]) layer-
        // propagator steps = evaluations x finite layers of the model (the flop model of SURVEY.md 8(d))
        unsigned long long tot = (r == 0) ? S.evals : 0u, lps = tot * (unsigned long long)(valid ? mmax - 1 : 0);
        for (int off = 32; off > 0; off >>= 1) {
            tot += __shfl_xor(tot, off);
            lps += __shfl_xor(lps, off);
        }
        if (lane == 0) {
            atomicAdd(A.neval, tot);
            atomicAdd(A.neval + (IFUNC == 2 ? 8 : 9), tot);
            atomicAdd(A.neval + (IFUNC == 2 ? 10 : 11), lps);
        }
    }
}

// ---- processing order: models by layer count, deepest first (counting sort, one workgroup) -------------
__global__ __launch_bounds__(1024) void order_kernel(int B, const int32_t *nlay, int32_t *perm, int Lcut, int32_t *split)
{
    __shared__ int bin[BH_MAX_LAYERS + 2];
    const int tid = threadIdx.x;
    for (int i = tid; i < BH_MAX_LAYERS + 2; i += 1024) bin[i] = 0;
    __syncthreads();
    for (int b = tid; b < B; b += 1024) {
        int n = nlay[b];
        n = n < 0 ? 0 : (n > BH_MAX_LAYERS + 1 ? BH_MAX_LAYERS + 1 : n);
        atomicAdd(&bin[n], 1);
    }
    __syncthreads();
    if (tid == 0) { // start offset of every depth, deepest first
        int acc = 0;
        for (int n = BH_MAX_LAYERS + 1; n >= 0; --n) {
            const int c = bin[n];
            bin[n] = acc;
            acc += c;
            if (n == Lcut + 1 && split != nullptr) split[0] = acc; // models with more than Lcut layers come first
        }
        if (split != nullptr && Lcut + 1 > BH_MAX_LAYERS + 1) split[0] = 0;
    }
    __syncthreads();
    for (int b = tid; b < B; b += 1024) {
        int n = nlay[b];
        n = n < 0 ? 0 : (n > BH_MAX_LAYERS + 1 ? BH_MAX_LAYERS + 1 : n);
        perm[atomicAdd(&bin[n], 1)] = b; // order inside a depth is arbitrary: per-model results do not depend on it
    }
}

// ---- processing order by predicted search length, paired over the SIMDs (one workgroup) ----------------------------
// Predicted length of a model's root search: the range of its S velocities (the search walks from the phase velocity of
// one period to that of the next in steps of dc: the longer the walk, the more secular evaluations; correlation with the
// evaluations counted by the oracle on bench.py's models: 0.94 Rayleigh, 0.81 Love).  Models sorted by it (bucket sort,
// 1024 buckets, longest first); wavefront `wid` of a target takes the group of rank slot_rank[wid] -- the ranks come
// from the launcher, which knows which wavefronts share a SIMD.  A batch of mixed depths is ordered by depth instead
// (deepest first, as order_kernel does): wavefronts of one depth matter more there.
constexpr int PAIR_BUCKETS = 1024;
constexpr int PAIR_MAX_B = 12288; // models the sorted list holds in LDS
__global__ __launch_bounds__(1024) void pair_order_kernel(int B, int Lmax, const int32_t *nlay, const double *vs, ptrdiff_t sl, ptrdiff_t sb,
                                                          int nt, PairOrderTarget t0, PairOrderTarget t1)
{
    __shared__ int bin[PAIR_BUCKETS];
    __shared__ int sorted[PAIR_MAX_B];
    __shared__ unsigned cmin_bits, cmax_bits;
    __shared__ int nmin, nmax;
    const int tid = threadIdx.x;
    for (int i = tid; i < PAIR_BUCKETS; i += 1024) bin[i] = 0;
    if (tid == 0) {
        cmin_bits = 0x7f800000u; // +inf
        cmax_bits = 0u;
        nmin = BH_MAX_LAYERS + 1;
        nmax = 0;
    }
    __syncthreads();
    // predicted cost (non-negative binary32: its bit pattern orders like the value) and the depth range of the batch;
    // reduced per thread and per wavefront first (1024 threads x 4 atomics on four LDS words serialise)
    constexpr int PER_THREAD = PAIR_MAX_B / 1024;
    float cost[PER_THREAD]; // of this thread's models b = tid + 1024 i (kept for the two bucket passes below)
    int depth[PER_THREAD];
    {
        unsigned tcmin = 0x7f800000u, tcmax = 0u;
        int tnmin = BH_MAX_LAYERS + 1, tnmax = 0;
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int b = tid + 1024 * i;
            cost[i] = 0.0f;
            depth[i] = 1;
            if (b >= B) continue;
            int n = nlay[b];
            n = n < 1 ? 1 : (n > Lmax ? Lmax : n);
            float lo = 1e30f, hi = -1e30f;
            for (int l = 0; l < n; ++l) {
                const float v = (float)vs[(ptrdiff_t)b * sb + (ptrdiff_t)l * sl];
                lo = v < lo ? v : lo;
                hi = v > hi ? v : hi;
            }
            float c = hi - lo;
            c = (c >= 0.0f && c < 1e30f) ? c : 0.0f; // (NaN / absurd models: anywhere)
            cost[i] = c;
            depth[i] = n;
            const unsigned cb = __float_as_uint(c);
            tcmin = cb < tcmin ? cb : tcmin;
            tcmax = cb > tcmax ? cb : tcmax;
            tnmin = n < tnmin ? n : tnmin;
            tnmax = n > tnmax ? n : tnmax;
        }
        for (int off = 32; off > 0; off >>= 1) {
            tcmin = min(tcmin, (unsigned)__shfl_xor((int)tcmin, off));
            tcmax = max(tcmax, (unsigned)__shfl_xor((int)tcmax, off));
            tnmin = min(tnmin, __shfl_xor(tnmin, off));
            tnmax = max(tnmax, __shfl_xor(tnmax, off));
        }
        if ((tid & (BH_WAVE - 1)) == 0) {
            atomicMin(&cmin_bits, tcmin);
            atomicMax(&cmax_bits, tcmax);
            atomicMin(&nmin, tnmin);
            atomicMax(&nmax, tnmax);
        }
    }
    __syncthreads();
    const bool ragged = nmin != nmax;
    const bool blocked = !ragged && t0.xcd > 0; // (the launcher checked the divisibilities)
    const int per = B / 8;                      // models of a block
    const float cmin = __uint_as_float(cmin_bits), cmax = __uint_as_float(cmax_bits);
    const float scale = (cmax > cmin) ? (float)(PAIR_BUCKETS - 1) / (cmax - cmin) : 0.0f;
    auto bucket = [&](int i) { // of this thread's i-th model
        if (ragged) return (nmax - depth[i]) < PAIR_BUCKETS ? (nmax - depth[i]) : PAIR_BUCKETS - 1; // deepest first
        int k = (int)((cmax - cost[i]) * scale); // longest first
        k = k < 0 ? 0 : (k > PAIR_BUCKETS - 1 ? PAIR_BUCKETS - 1 : k);
        if (blocked) k = ((tid + 1024 * i) / per) * (PAIR_BUCKETS / 8) + (k >> 3); // block-major: eight sorted lists one after the other
        return k;
    };
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i)
        if (tid + 1024 * i < B) atomicAdd(&bin[bucket(i)], 1);
    __syncthreads();
    // exclusive scan of the 1024 buckets, one per thread: inside the wavefronts by shuffles, then over the 16 wavefront sums
    {
        __shared__ int wsum[1024 / BH_WAVE];
        const int lane = tid & (BH_WAVE - 1), wv = tid / BH_WAVE;
        const int mine = bin[tid];
        int acc = mine;
        for (int off = 1; off < BH_WAVE; off <<= 1) {
            const int o = __shfl_up(acc, off);
            if (lane >= off) acc += o;
        }
        if (lane == BH_WAVE - 1) wsum[wv] = acc;
        __syncthreads();
        if (tid < 1024 / BH_WAVE) {
            int w = wsum[tid];
            for (int off = 1; off < 1024 / BH_WAVE; off <<= 1) {
                const int o = __shfl_up(w, off);
                if (tid >= off) w += o;
            }
            wsum[tid] = w;
        }
        __syncthreads();
        bin[tid] = (wv > 0 ? wsum[wv - 1] : 0) + acc - mine; // start offset of the bucket
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i)
        if (tid + 1024 * i < B) sorted[atomicAdd(&bin[bucket(i)], 1)] = tid + 1024 * i; // (order inside a bucket is arbitrary)
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const PairOrderTarget T = t == 0 ? t0 : t1;
        for (int pos = tid; pos < B; pos += 1024) {
            const int wid = pos / T.mpw, m = pos - wid * T.mpw;
            int src = pos; // the last (possibly partly filled) wavefront and mixed-depth batches: sorted order as it is
            if (!ragged && T.slot_rank != nullptr && wid < T.nwaves - 1) src = T.slot_rank[wid] * T.mpw + m;
            if (blocked) { // wavefront wid runs on XCD x as that XCD's q-th wavefront of the target: the q-th group of block x
                const int x = (T.xcd == 2) ? (wid >> 1) & 7 : (wid >> 2) & 7;
                const int q = (T.xcd == 2) ? (((wid >> 4) << 1) | (wid & 1)) : (((wid >> 5) << 2) | (wid & 3));
                src = x * per + q * T.mpw + m;
            }
            T.perm[pos] = sorted[src];
        }
    }
}

// log / powf of the host's libm (what the reference's compiled Fortran calls), restated in bh_libm.h;
// arguments outside the restated paths (never produced by a physical model) use the device library.
__device__ __forceinline__ double sphere_log(double x)
{
    double y;
    return bhp_log(x, &y, bhp_log_data) ? y : log(x);
}
__device__ __forceinline__ float sphere_powf(float x, float y)
{
    float r;
    return bhp_powf(x, y, &r, bhp_powf_log2_data, bhp_exp2f_data) ? r : powf(x, y);
}

// ---- earth flattening, surfdisp96.f:486-553 (`sphere`, both calls) --------------------------------
// One lane per model.  Arithmetic widths as in the Fortran (model arrays binary32, radii binary64);
// log() and powf() are the glibc restatements, so the transformed model is bit-identical to the reference's.
__global__ void sphere_kernel(int B, int Lmax, const int32_t *nlay, const double *h, const double *vp,
                              const double *vs, const double *rho, ptrdiff_t sl, ptrdiff_t sb, double *oh,
                              double *ovp, double *ovs, double *orl, double *orr)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int mmax = nlay[b];
    const double ar = 6370.0;
    double dr = 0.0, r0 = ar;
    for (int i = 0; i < mmax; ++i) {
        const ptrdiff_t o = (ptrdiff_t)b * sb + (ptrdiff_t)i * sl;
        const float d = (i == mmax - 1) ? 1.0f : (float)h[o]; // d(mmax) = 1.0 while transforming
        const float a = (float)vp[o], bb = (float)vs[o], rt = (float)rho[o];
        dr = dr + (double)d;
        const double r1 = ar - dr;
        const double z0 = ar * sphere_log(ar / r0);
        const double z1 = ar * sphere_log(ar / r1);
        const float dn = (i == mmax - 1) ? 0.0f : (float)(z1 - z0); // d(mmax) = 0 afterwards
        const double tmp = (ar + ar) / (r0 + r1);                   // layer mid-point
        const float btp = (float)tmp;
        float p5 = btp * btp; // btp**(-5) the way compiler-rt's __powisf2 does it
        p5 = p5 * p5;
        p5 = btp * p5;
        const size_t q = (size_t)i * B + b;
        oh[q] = (double)dn;
        ovp[q] = (double)(float)((double)a * tmp);
        ovs[q] = (double)(float)((double)bb * tmp);
        orl[q] = (double)(rt * (1.0f / p5));
        orr[q] = (double)(rt * sphere_powf(btp, -2.275f));
        r0 = r1;
    }
}

// ---- np.interp, row-wise (numpy/core/src/multiarray/compiled_base.c: arr_interp) -------------------
__global__ void interp_kernel(int B, int K0, const double *x0, const double *y0, int ld0, int K1,
                              const double *x1, double *y1, int ld1)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * K1) return;
    const int b = t / K1, k = t % K1;
    const double x = x1[k];
    const double *fp = y0 + (size_t)b * ld0;
    double r;
    if (x < x0[0]) r = fp[0];
    else if (x > x0[K0 - 1]) r = fp[K0 - 1];
    else {
        int j = 0; // largest j with x0[j] <= x
        for (int lo = 0, hi = K0; lo < hi;) {
            const int mid = (lo + hi) >> 1;
            if (x0[mid] <= x) { j = mid; lo = mid + 1; } else hi = mid;
        }
        if (j == K0 - 1 || x0[j] == x) r = fp[j];
        else {
            const double slope = (fp[j + 1] - fp[j]) / (x0[j + 1] - x0[j]);
            r = slope * (x - x0[j]) + fp[j];
            if (r != r) { // numpy's nan rescue (an infinity in fp)
                r = slope * (x - x0[j + 1]) + fp[j + 1];
                if (r != r && fp[j] == fp[j + 1]) r = fp[j];
            }
        }
    }
    y1[(size_t)b * ld1 + k] = r;
}

} // namespace


void bh_launch_order(int B, const int32_t *nlay, int32_t *perm, int Lcut, int32_t *split, hipStream_t stream)
{
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, stream, B, nlay, perm, Lcut, split);
}

void bh_launch_pair_order(int B, int Lmax, const int32_t *nlay, const double *vs, ptrdiff_t sl, ptrdiff_t sb, int nt,
                          const PairOrderTarget *tg, hipStream_t stream)
{
    hipLaunchKernelGGL(pair_order_kernel, dim3(1), dim3(1024), 0, stream, B, Lmax, nlay, vs, sl, sb, nt, tg[0], tg[nt > 1 ? 1 : 0]);
}
bool bh_pair_order_fits(int B) { return B <= PAIR_MAX_B; }

void bh_launch_sphere(int B, int Lmax, const int32_t *nlay, const double *h, const double *vp,
                      const double *vs, const double *rho, ptrdiff_t sl, ptrdiff_t sb, double *oh,
                      double *ovp, double *ovs, double *orho_love, double *orho_ray, hipStream_t stream)
{
    hipLaunchKernelGGL(sphere_kernel, dim3((B + 127) / 128), dim3(128), 0, stream, B, Lmax, nlay, h, vp, vs, rho,
                       sl, sb, oh, ovp, ovs, orho_love, orho_ray);
}

void bh_launch_interp(int B, int K0, const double *x0, const double *y0, int ld0, int K1,
                      const double *x1, double *y1, int ld1, hipStream_t stream)
{
    const int n = B * K1;
    hipLaunchKernelGGL(interp_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, B, K0, x0, y0, ld0, K1, x1, y1, ld1);
}

size_t bh_swd_lds_bytes(int Lmax, int K, int mode) { return LANE_TAB_PAD + lane_wave_bytes(Lmax, K, mode); } // one wavefront per workgroup

// doubles of the global work array of a launch (Neville orders from NEV_LO on, one column per lane)
size_t bh_swd_nev_high_doubles(int B, int look)
{
    int J = 1;
    while (2 * J <= look && 2 * J <= 16) J *= 2;
    const int mpw = BH_WAVE / J;
    const size_t waves = (size_t)(((B + mpw - 1) / mpw + 1) / 2) * 2; // (rounded up to a whole workgroup of two)
    return (size_t)2 * (NEV_MAX - NEV_LO) * waves * BH_WAVE;
}

void bh_launch_swd(const SwdKernelArgs &a, int iwave, hipStream_t stream)
{
    SwdKernelArgs b = a;
    int J = 1;
    while (2 * J <= a.look && 2 * J <= 16) J *= 2; // largest power of two <= look
    b.look = J;
    const int no_fair = bh_tuning().swd_no_fair ? 1 : 0; // (experiment switches, bh_tuning.h)
    const int slice = bh_tuning().swd_slice;
    b.fair = (no_fair || a.fair < 0) ? 0 : (slice > 0 ? slice : (a.fair > 0 ? a.fair : 16));
    const int mpw = BH_WAVE / J;
    const int waves = (a.B + mpw - 1) / mpw;
    // two wavefronts per workgroup once the call has more wavefronts than fit with one (a.fair carries the engine's
    // count class: 12 = more than 2048 wavefronts in the call)
    const bool two = a.fair == 12 && LANE_TAB_PAD + 2 * lane_wave_bytes(a.Lmax, a.K, a.mode) <= 64 * 1024;
    const size_t wb = lane_wave_bytes(a.Lmax, a.K, a.mode);
    const size_t lds = LANE_TAB_PAD + (two ? 2 : 1) * wb;
    const dim3 grid(two ? (waves + 1) / 2 : waves), block((two ? 2 : 1) * BH_WAVE);
    // (wave type, look-ahead, wavefronts per workgroup, refinement) -> instantiation
    const bool no_simple = bh_tuning().swd_no_simple != 0;
    // Measured (c2 batches, with / without): reference sequence -3 % at B = 16 384 (4 trial lanes per model) but +4 % at
    // 65 536 and +6 % at 131 072 (one lane per model: the Love build drops to 165 registers there, a third wavefront per SIMD
    // upsets the Rayleigh / Love pairs the time-sliced priorities are tuned for); short refinement -3 % at 65 536.
    const bool simple = a.igr == 0 && a.mode <= 1 && !no_simple && (J > 1 || a.fast);
#define BH_LANE_LAUNCH(IF, LK, WP, FS, SI) hipLaunchKernelGGL((swd_kernel<IF, LK, WP, FS, SI>), grid, block, lds, stream, b)
#define BH_LANE_PICK_FS(IF, LK, WP)                                                  \
    do {                                                                             \
        if (a.fast && a.igr == 0) {                                                  \
            if (simple && a.farith) hipLaunchKernelGGL((swd_kernel<IF, LK, WP, 2, true, true>), grid, block, lds, stream, b); \
            else if (simple) BH_LANE_LAUNCH(IF, LK, WP, 2, true);                    \
            else BH_LANE_LAUNCH(IF, LK, WP, 2, false);                               \
        } else if (simple) BH_LANE_LAUNCH(IF, LK, WP, 0, true);                      \
        else BH_LANE_LAUNCH(IF, LK, WP, 0, false);                                   \
    } while (0)
#define BH_LANE_PICK_WP(IF, LK) do { if (two) BH_LANE_PICK_FS(IF, LK, 2); else BH_LANE_PICK_FS(IF, LK, 1); } while (0)
#define BH_LANE_PICK_LK(IF) do { if (J > 1) BH_LANE_PICK_WP(IF, true); else BH_LANE_PICK_WP(IF, false); } while (0)
    if (iwave == 1) BH_LANE_PICK_LK(1);
    else BH_LANE_PICK_LK(2);
#undef BH_LANE_PICK_LK
#undef BH_LANE_PICK_WP
#undef BH_LANE_PICK_FS
#undef BH_LANE_LAUNCH
}

// ---- launch plan ------------------------------------------------------------------------------------
// Which mapping (one lane per model, or G lanes per model) and how many trial velocities per round and
// target.  Cost model, calibrated on MI355X with 10-layer models and 30 periods (profiles/, docs/HISTORY.md 3.1):
// at this kernel's register budget 2 wavefronts are resident per SIMD = 2048 on the chip; a launch runs in
// ceil(wavefronts / 2048) rounds, each as long as its longest wavefront.  Relative wavefront durations:
//   group kernel, Rayleigh: 1 / .64 / .52 / .45 / .36 for 1 / 2 / 3 / 4 / 7 trials per round,
//   group kernel, Love:     .735 / .54 / .39 / .33 / .26 (with two in-group trials at the first two levels),
//   one lane per model:     3.44 (64 models per wavefront, every layer term serial; the targets of a call side by side);
//                           x 1 / .60 / .40 / .30 / .26 with 1 / 2 / 4 / 8 / 16 trial lanes per model.
// More trials shorten every model's chain of dependent secular evaluations (what a small batch is bound
// by) but cost lanes, i.e. wavefronts.  The plan minimises rounds x longest wavefront.
namespace {
constexpr int PLAN_LEVELS = 5;
const int plan_trials[PLAN_LEVELS] = {1, 2, 3, 4, 7};
const double plan_dur[2][PLAN_LEVELS] = {{0.735, 0.54, 0.39, 0.33, 0.26}, {1.0, 0.64, 0.52, 0.45, 0.36}};
constexpr int PLAN_SLOTS = 2048;
constexpr double PLAN_LANE_PER_MODEL = 3.44; // in units of a 1-trial Rayleigh wavefront of the group kernel sharing its SIMD
constexpr int PLAN_LANE_LEVELS = 5; // 1, 2, 4, 8, 16 trial lanes per model
const double plan_lane_dur[PLAN_LANE_LEVELS] = {1.0, 0.60, 0.40, 0.30, 0.26};

int plan_fit(int G, int level) // largest trial count <= the level's that fits a wavefront
{
    int J = plan_trials[level];
    while (J > 1 && G * J > BH_WAVE) --J;
    return J;
}
} // namespace

int bh_swd_pick_group(int B, int ntargets, int Lmax)
{
    // one lane per finite layer (at least 5: the width of the Rayleigh vector recursion), capped at
    // 16, so that phase A needs a single round
    int G = Lmax - 1;
    if (G < 5) G = 5;
    if (G > 16) G = 16;
    (void)B; (void)ntargets;
    return G;
}

// Returns the plan's cost; *G = 1 selects the lane-per-model kernel (look[] is then all 1).
// Gforce > 0: the caller fixed the lanes per model, only the trials are planned.
double bh_swd_plan(int B, int Lmax, int ntargets, const int *iwave, int Gforce, int *G, int *look)
{
    const int Gg = Gforce > 1 ? Gforce : bh_swd_pick_group(B, ntargets, Lmax);
    int lvl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, best_lvl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double best = 1e300;
    auto cost = [&](const int *l) {
        long waves = 0;
        double dmax = 0.0, dsum = 0.0;
        for (int t = 0; t < ntargets; ++t) {
            const int J = plan_fit(Gg, l[t]);
            const int mpw = BH_WAVE / (Gg * J);
            waves += (B + mpw - 1) / mpw;
            const double d = plan_dur[iwave[t] == 2 ? 1 : 0][l[t]];
            dmax = d > dmax ? d : dmax;
            dsum += d;
        }
        const long rounds = (waves + PLAN_SLOTS - 1) / PLAN_SLOTS;
        // (two wavefronts on a SIMD slow each other down a little; ties go to the shorter wavefronts)
        return (double)rounds * dmax * (waves > PLAN_SLOTS / 2 ? 1.15 : 1.0) + 1e-3 * dsum;
    };
    if (ntargets <= 4) { // exhaustive
        for (;;) {
            const double c = cost(lvl);
            if (c < best) {
                best = c;
                for (int t = 0; t < ntargets; ++t) best_lvl[t] = lvl[t];
            }
            int t = 0;
            while (t < ntargets && ++lvl[t] == PLAN_LEVELS) lvl[t++] = 0;
            if (t == ntargets) break;
        }
    } else { // same level for all targets
        for (int l = 0; l < PLAN_LEVELS; ++l) {
            for (int t = 0; t < ntargets; ++t) lvl[t] = l;
            const double c = cost(lvl);
            if (c < best) {
                best = c;
                for (int t = 0; t < ntargets; ++t) best_lvl[t] = l;
            }
        }
    }
    // the lane-per-evaluation kernel: one launch per target, 64 / J models per wavefront, J = trial lanes per model.
    // Measured (B = 4096 ... 32 768, Rayleigh + Love side by side): one wavefront per SIMD or less -> 14.9 / 8.9 / 5.9 /
    // 4.4 / 3.9 ms for J = 1 / 2 / 4 / 8 / 16; the factor for shared SIMDs below.
    double c1 = 1e300;
    int J1 = 1;
    bool any_rayleigh = false;
    for (int t = 0; t < ntargets; ++t) any_rayleigh = any_rayleigh || iwave[t] == 2;
    for (int l = 0; l < PLAN_LANE_LEVELS; ++l) {
        const int J = 1 << l;
        const int mpw = BH_WAVE / J;
        const long w = (long)ntargets * ((B + mpw - 1) / mpw);
        double c = PLAN_LANE_PER_MODEL * plan_lane_dur[l] * (any_rayleigh ? 1.0 : 0.85);
        // wavefronts sharing a SIMD (time-sliced priorities): x 1.25 as soon as some do, x 1.4 when all do, then
        // (wavefronts / 2048)^0.85
        if (w > PLAN_SLOTS) c *= 1.4 * std::pow((double)w / PLAN_SLOTS, 0.85);
        else if (w > PLAN_SLOTS / 2) c *= 1.25 + 0.15 * (double)(w - PLAN_SLOTS / 2) / (PLAN_SLOTS / 2);
        if (c < c1) {
            c1 = c;
            J1 = J;
        }
    }
    if (Gforce == 1 || (Gforce <= 0 && c1 < best)) {
        *G = 1;
        for (int t = 0; t < ntargets; ++t) look[t] = J1;
        return c1;
    }
    *G = Gg;
    for (int t = 0; t < ntargets; ++t) look[t] = plan_fit(Gg, best_lvl[t]);
    return best;
}
