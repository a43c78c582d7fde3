// bayhunter_amd/csrc/swd_group_kernel.hip -- Rayleigh/Love dispersion with G lanes per model (small batches).
// Same search, same bits as swd_kernel.hip (one lane per model); see the block comment at the kernel.
// Built with -mllvm -disable-machine-licm (Makefile): the round loop carries ~60 distinct f64 constants of the
// glibc-exact sincos / exp; hoisted out of the loop they cost more registers than the kernel has.
#include "../../include/bh_engine_debug.h"
#include "bh_device.h"
#include "bh_tuning.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>
#define BH_HD __device__ __forceinline__
#define BH_TAB static __device__ const
#include "bh_libm.h"

namespace {
#include "swd_common.h"

// =================================================================================================
// Kernel 2: G lanes = one model (64/G models per wavefront), G chosen at launch time.
// For batches that cannot fill the chip with one lane per model (B = 4096 gives only 64
// wavefronts for 1024 SIMDs) the work of ONE secular evaluation is spread over the G lanes of
// the model's group:
//   phase A  lane li computes the layer terms of layers li, li+G, ... (the transcendental-heavy
//            part: sqrt, sin/cos or exp, the compound-matrix entries) and parks them in LDS;
//   phase B  the strictly sequential bottom-up recursion over the parked layers.
//            Rayleigh, G >= 5: lane li owns component (li mod 5) of the 5-vector: it forms
//            ee(i) = sum_j e(j)*ca(j,i) from column i of the parked matrix, the five values are
//            exchanged through LDS, every lane takes the max-norm, divides its own component and
//            the normalised vector is exchanged again.  Otherwise (Love, or G < 5) every lane of
//            the group runs the whole recursion redundantly.
//            Either way all lanes of a group end up with the same secular value and step the
//            same search state; no broadcast is needed.
// Every floating-point operation and its order are those of kernel 1: the kernels return
// identical bits.  All dispersion targets of a call go into one launch (blockIdx.y = target),
// so Rayleigh and Love wavefronts share the chip.
// A workgroup is GROUP_WPB independent wavefronts that only share the LDS copy of the libm tables;
// after start-up there is no barrier: LDS operations of a wavefront execute in order, wave_sync()
// only pins the compiler's ordering.  Look-ahead, Love's in-group trials, the processing order and the
// two depth classes of ragged batches are described at their code.
// =================================================================================================
constexpr int CA_STRIDE = 26;
constexpr int LOVE_TERMS = 6; // doubles per parked Love layer and trial (5 used): up to 4 trials share a row

// Phase B of the group kernel, Rayleigh.  `cam` = this model's parked layers (column-major
// 5x5 each), e = half-space vector on entry / surface vector on exit.
//   PAR5   lane owns component `col`: one dot product per layer, the five results are exchanged
//          with ds_bpermute (faster than an LDS write/read round trip: 56 vs 116 cycles) and every
//          lane normalises all five itself -- one exchange per layer, no branch.
//   !PAR5  every lane runs the full 5x5 product (used for G < 5 and for the exact re-run).
//   RAGGED the wavefront holds models with different layer counts (or a water layer): layers a
//          model does not have are masked with selects; the uniform case has no masking at all.
template <bool PAR5, bool RAGGED, bool EXACT, bool FA = false>
__device__ __forceinline__ void rayleigh_chain_group(double e[5], const double *cam, int col,
                                                     int gbase, int mtop, int mmax, int llw,
                                                     DivRange &dr)
{
    // uniform wavefronts: the layer count is the same in every lane -> scalar loop control
    const int mstart = (RAGGED ? mtop : __builtin_amdgcn_readfirstlane(mmax)) - 2;
    if (PAR5) {
        // software-pipelined: the column of layer m-1 is fetched while layer m is exchanged
        const double *cc = cam + (size_t)(mstart > 0 ? mstart : 0) * CA_STRIDE + 5 * col;
        double c0 = cc[0], c1 = cc[1], c2 = cc[2], c3 = cc[3], c4 = cc[4];
#pragma unroll 3
        for (int m = mstart; m >= 0; --m) {
            const bool on = !RAGGED || (m <= mmax - 2 && m >= llw - 1);
            const double *cn = cam + (size_t)(m > 0 ? m - 1 : 0) * CA_STRIDE + 5 * col;
            double n0 = cn[0], n1 = cn[1], n2 = cn[2], n3 = cn[3], n4 = cn[4];
            double ee = 0.0;
            if (FA) { // (fast arithmetic: two short chains of fused multiply-adds)
                ee = __builtin_fma(e[4], c4, __builtin_fma(e[2], c2, e[0] * c0)) + __builtin_fma(e[3], c3, e[1] * c1);
            } else {
                ee = ee + e[0] * c0;
                ee = ee + e[1] * c1;
                ee = ee + e[2] * c2;
                ee = ee + e[3] * c3;
                ee = ee + e[4] * c4;
            }
            const double v0 = __shfl(ee, gbase + 0), v1 = __shfl(ee, gbase + 1), v2 = __shfl(ee, gbase + 2),
                         v3 = __shfl(ee, gbase + 3), v4 = __shfl(ee, gbase + 4);
            // keep the fetch of layer m-1 in THIS iteration (the compiler otherwise sinks it to the top of the next
            // one and waits for it there: ~100 cycles of the ~600 a layer takes); LDS returns in order, so by the
            // time the exchange above has arrived these have as well
            asm volatile("" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4));
            double en[5];
            if (FA) {
                fa::normalize5(v0, v1, v2, v3, v4, en);
            } else {
                DivRange d2 = dr;
                normalize5<EXACT>(v0, v1, v2, v3, v4, en, d2);
                if (on) dr = d2;
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) e[i] = on ? en[i] : e[i];
            c0 = n0; c1 = n1; c2 = n2; c3 = n3; c4 = n4;
        }
    } else {
        for (int m = mstart; m >= 0; --m) {
            const bool on = !RAGGED || (m <= mmax - 2 && m >= llw - 1);
            const double *cc = cam + (size_t)m * CA_STRIDE;
            double ee[5], en[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                double acc = 0.0;
                if (FA) {
                    acc = __builtin_fma(e[4], cc[5 * i + 4], __builtin_fma(e[2], cc[5 * i + 2], e[0] * cc[5 * i])) +
                          __builtin_fma(e[3], cc[5 * i + 3], e[1] * cc[5 * i + 1]);
                } else {
#pragma unroll
                    for (int j = 0; j < 5; ++j) acc = acc + e[j] * cc[5 * i + j];
                }
                ee[i] = acc;
            }
            if (FA) {
                fa::normalize5(ee[0], ee[1], ee[2], ee[3], ee[4], en);
            } else {
                DivRange d2 = dr;
                normalize5<EXACT>(ee[0], ee[1], ee[2], ee[3], ee[4], en, d2);
                if (on) dr = d2;
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) e[i] = on ? en[i] : e[i];
        }
    }
}

// Phase B of the group kernel, Love: parked per layer (cosq, y, z, xmu, rcp(xmu)).
template <bool RAGGED, bool EXACT, bool COUNT, bool FA = false>
__device__ __forceinline__ void love_chain_group(double &e1, double &e2, const double *cam, int mtop,
                                                 int mmax, int llw, DivRange &dr, LoveCount &lc)
{
    const int mstart = (RAGGED ? mtop : __builtin_amdgcn_readfirstlane(mmax)) - 2;
    // software-pipelined: the terms of layer m-1 are fetched while layer m is processed
    const double2 *src = reinterpret_cast<const double2 *>(cam + (size_t)(mstart > 0 ? mstart : 0) * CA_STRIDE);
    double2 p0 = src[0], p1 = src[1], p2 = src[2];
#pragma unroll 3
    for (int m = mstart; m >= 0; --m) {
        const bool on = !RAGGED || (m <= mmax - 2 && m >= llw - 1);
        const double2 *nx = reinterpret_cast<const double2 *>(cam + (size_t)(m > 0 ? m - 1 : 0) * CA_STRIDE);
        const double2 q0 = nx[0], q1 = nx[1], q2 = nx[2];
        double n1 = e1, n2 = e2;
        DivRange d2 = dr;
        if (FA) fa::love_step(n1, n2, p0.x, p0.y, p1.x);
        else love_step<EXACT>(n1, n2, p0.x, p0.y, p1.x, p1.y, p2.x, d2);
        if (on) {
            if (COUNT) lc.layer(p2.y, e2, n2); // (p2.y: floor(q / pi) of the layer, parked by phase A)
            e1 = n1;
            e2 = n2;
            dr = d2;
        }
        p0 = q0; p1 = q1; p2 = q2;
    }
}
 // doubles per parked layer: 25 (Rayleigh, column-major 5x5) / 4 (Love)

__device__ __forceinline__ void park_ca25(double *dst, const Ca19 &c)
{
    // column i (0-based) at dst[5*i + j] = ca(j+1, i+1)
    const double ca11 = c.c[0], ca12 = c.c[1], ca13 = c.c[2], ca14 = c.c[3], ca15 = c.c[4];
    const double ca21 = c.c[5], ca23 = c.c[6], ca24 = c.c[7], ca22 = c.c[8];
    const double ca41 = c.c[9], ca42 = c.c[10], ca43 = c.c[11], ca51 = c.c[12], ca53 = c.c[13];
    const double ca31 = c.c[14], ca32 = c.c[15], ca33 = c.c[16], ca34 = c.c[17], ca35 = c.c[18];
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    d2[0] = make_double2(ca11, ca21);  d2[1] = make_double2(ca31, ca41);   // col 1: 11 21 31 41 51
    d2[2] = make_double2(ca51, ca12);  d2[3] = make_double2(ca22, ca32);   // col 2: 12 22 32 42 52
    d2[4] = make_double2(ca42, ca41);  d2[5] = make_double2(ca13, ca23);   // (ca52 = ca41) col 3: 13 23 33 43 53
    d2[6] = make_double2(ca33, ca43);  d2[7] = make_double2(ca53, ca14);   // col 4: 14 24 34 44 54
    d2[8] = make_double2(ca24, ca34);  d2[9] = make_double2(ca22, ca21);   // (ca44 = ca22, ca54 = ca21)
    d2[10] = make_double2(ca15, ca14); d2[11] = make_double2(ca35, ca12);  // col 5: 15 25 35 45 55
    d2[12] = make_double2(ca11, 0.0);                                      // (ca25=ca14, ca45=ca12, ca55=ca11)
}

// Wavefronts per workgroup: they are independent (no barrier after start-up) and only share one LDS
// copy of the libm tables, which is what lets 8 wavefronts fit a CU's 160 KB of LDS.
constexpr int GROUP_WPB = 2;
constexpr unsigned PRIO_BLOCK = 8; // rounds between priority decisions (power of two)
constexpr int LIBM_TAB_PAD = (LIBM_TAB_BYTES + 15) & ~15;

// LDS ordering inside ONE wavefront: its LDS instructions execute in order, so a write by one lane is
// visible to a later read by another lane of the same wavefront; only the compiler must not reorder.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// WPB = wavefronts per workgroup (they only share the libm tables): 2 by default; 4 when receiver-function workgroups
// are to run beside the kernel -- a CU's eight wavefronts then hold two copies of the tables instead of four, which is
// what leaves a CU's LDS room for one RF workgroup (bh_engine.hip: co-resident receiver function).
// FAST: the build with the short refinement (SearchT<.., FAST>, swd_common.h) for the phase-velocity targets.
// PROF: the build with the evaluation counters, phase clocks and wavefront trace (launches with A.neval set).
// ADAPT: a launch of ONE model per wavefront (a chain window, a single model): every wavefront sizes its lane groups and its
// trials per round for its OWN model -- a lane per finite layer (8 ... 16), as many trials as lanes and the wavefront's LDS
// region admit (at most ADAPT_MAX_TRIALS), LDS rows for the model's own layers instead of the arrays' capacity.  With the
// capacity's rows (21 for the chains) four trials fit; a window's dispersion time falls with every further trial
// (4 / 6 / 8 trials: 1.82 / 1.53 / 1.42 ms for 1016 models of 3-9 layers), so the shallow models of a window get eight,
// its deep ones keep four.  Scheduling only: which values the search consumes does not depend on it.
constexpr int ADAPT_MAX_TRIALS = 8;
// CNTB: the build with the counted scan of Love targets (SearchT, swd_common.h: same brackets, same bits, a third of the scan's
// evaluations).  Its state machine and the mode count in Love's recursion cost every wavefront of the launch a little (code
// size, scalar registers), so it is compiled into the launches where it pays -- one model per wavefront (ADAPT: a single
// model, the chains' windows), launches of Love targets only -- and not where Rayleigh wavefronts set the time anyway
// (c2: Rayleigh + Love at B = 4096: 3.37 ms without, 3.47 with; the Rayleigh wavefront alone on its SIMD takes 3.06).
// FA: the build with the fast arithmetic (swd_fa.h; bh_engine_set_swd_arith): launches in which every target takes the short
// refinement (FASTM = 2) -- tolerance-level parity, a guard for signs the rounding error could decide.
// (the FA builds are compiled in a translation unit of their own, swd_group_fa.hip, which includes this file: own flags)
#ifndef BH_GROUP_WAVES
#define BH_GROUP_WAVES 2
#endif
template <int WPB, int FASTM, bool SIMPLE, bool PROF, bool ADAPT, bool CNTB, bool FA = false>
__global__ __launch_bounds__(BH_WAVE * WPB) __attribute__((amdgpu_waves_per_eu(BH_GROUP_WAVES, BH_GROUP_WAVES))) void swd_group_kernel(SwdMultiArgs A, int Gflags, int wave_lds)
{
    // "this workgroup is resident": what a second stream waits for before it dispatches wavefronts beside these
    if (A.started != nullptr && threadIdx.x == 0) atomicAdd(A.started, 1u);
    const int cls = (A.split != nullptr) ? (int)blockIdx.z : 1; // 0 = the deep models of a ragged batch, 1 = the rest
    // Wavefront -> (target, index).  Two targets in a one-dimensional grid are INTERLEAVED wavefront by wavefront in
    // proportion to their wavefront counts, so that every CU gets its share of both: dispatched target after
    // target, the Rayleigh wavefronts fill whole CUs before the first Love wavefront starts, and a CU of eight
    // Rayleigh wavefronts runs them up to 30 % slower than a mixed one (LDS traffic: 150 LDS instructions per
    // Rayleigh round, 61 per Love round).  A workgroup's wavefronts only share the libm tables.
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / BH_WAVE));
    // PASSES (the re-run of the models a guard listed, one model per wavefront, reference sequence): the launch is a SMALL grid
    // (bh_launch_swd_group: bh_tuning.h swd_rerun_wgs = 256 per target) whose workgroups stride over the list, whose length lives on
    // the device -- nearly always it is empty or a handful of models, and a launch sized for the worst case (one wavefront per
    // model of the batch: 2048 workgroups at B = 4096) had to be dispatched workgroup by workgroup only to leave at once (4.5 us
    // alone, 0.2 ms when receiver-function workgroups of the other stream were waiting for the same wave slots).
    constexpr bool PASSES = ADAPT && FASTM == 0;
    int pass = 0;
next_pass:
    const int wg_x = (int)blockIdx.x + ((PASSES && A.rerun != 0) ? pass * (int)gridDim.x : 0);
    int wid = wg_x * WPB + wave, ty = (int)blockIdx.y;
    bool beyond = false; // (interleaved grid: a last, odd wavefront may have nothing to do)
    if (A.wg_n1 > 0) {
        const long long n1 = A.wg_n1, N = (long long)A.wg_n0 + n1;
        beyond = wid >= (int)N;
        const int l0 = (int)(((long long)wid * n1) / N), l1 = (int)((((long long)wid + 1) * n1) / N);
        ty = (l1 > l0) ? 1 : 0;
        wid = (l1 > l0) ? l0 : wid - l0;
    }
    int G = A.lanes[cls];
    const SwdTarget T = A.t[ty];
    int nlist = A.B; // entries of the processing order this launch covers
    if (T.count != nullptr) { // (a re-run of listed models: the count lives on the device)
        const int c = *T.count;
        nlist = c < nlist ? c : nlist;
    }
    int J = T.look > 1 ? T.look : 1; // look-ahead: trial velocities per round (per target), one lane group each
    int rows_own = 0;
    if (ADAPT) { // (one class, one model per wavefront: wavefront `wid` has model `wid` of the processing order)
        const int32_t *perm0 = T.perm != nullptr ? T.perm : A.perm;
        const bool v0 = !beyond && wid < nlist;
        int m0 = v0 ? A.nlay[perm0 ? perm0[wid] : wid] : 2;
        m0 = __builtin_amdgcn_readfirstlane(m0);
        m0 = m0 < 2 ? 2 : (m0 > A.rows[1] ? A.rows[1] : m0);
        const int fin = m0 - 1;
        G = fin < 8 ? 8 : (fin > 16 ? 16 : fin);
        const int fixed = (2 * NEV_MAX + ((T.K + 1) & ~1)) * (int)sizeof(double) + ((4 * m0 * (int)sizeof(float) + 15) & ~15) + 64;
        const int jl = (wave_lds - fixed) / (fin * CA_STRIDE * (int)sizeof(double));
        J = BH_WAVE / G;
        J = J > jl ? jl : J;
        J = J > ADAPT_MAX_TRIALS ? ADAPT_MAX_TRIALS : J;
        J = J < 1 ? 1 : J;
        rows_own = m0;
    }
    while (J > 1 && G * J > BH_WAVE) --J;
    // Love only: further trials INSIDE a lane group.  Its recursion is scalar (every lane of the group
    // would repeat it), so lane l runs trial l mod JL instead; only the layer terms cost JL passes.
    const int JL = (T.iwave == 1 && T.inlook > 1) ? T.inlook : 1;
    const int LPM = G * J;         // lanes per model: J groups of G lanes, group r evaluates candidate r
    const int MPW = ADAPT ? 1 : BH_WAVE / LPM; // models per wavefront (lanes >= MPW*LPM idle along as clones of lane 0)
    extern __shared__ __align__(16) unsigned char smem_all[];
    const int lane = threadIdx.x & (BH_WAVE - 1);
    // the workgroup's shared copy of the libm tables, then one private region per wavefront
    // this launch's range of the processing order (see SwdMultiArgs::split)
    int lo = 0, hi = nlist;
    if (A.split != nullptr) {
        const int ndeep = A.split[0];
        if (cls == 0) hi = ndeep;
        else lo = ndeep;
    }
    if (A.wg_n1 == 0 && lo + wg_x * WPB * MPW >= hi) return; // whole workgroup beyond the range (grid = worst case)
    const LibmTabs LT = stage_libm_tables(smem_all, threadIdx.x, BH_WAVE * WPB); // (a later pass writes the same values again)
    if (pass == 0) __syncthreads();
    if (beyond || lo + wid * MPW >= hi) return;
    unsigned char *smem = smem_all + LIBM_TAB_PAD + (size_t)wave * wave_lds;
    const bool spare = lane >= MPW * LPM;
    const int g = spare ? 0 : lane / LPM;        // model slot inside the wave
    const int rr = spare ? 0 : (lane % LPM) / G; // which candidate this lane's group evaluates
    const int li = spare ? 0 : lane % G;         // this lane's index inside its group
    const int slot = g * J + rr;                 // group index inside the wave
    const int sidx = lo + wid * MPW + g; // position in the processing order
    const bool valid = sidx < hi;
    const int32_t *perm = T.perm != nullptr ? T.perm : A.perm; // processing order: the target's own (SIMD pairing) or the batch's
    const int ib = valid ? (perm ? perm[sidx] : sidx) : 0;
    const int Lmax = ADAPT ? rows_own : A.rows[cls]; // LDS rows per model of this class (>= every layer count it meets)
    const int K = T.K;
    const int ifunc = T.iwave; // 1 Love, 2 Rayleigh: uniform per wavefront

    // LDS carve-up of the wavefront's region (all offsets multiples of 16 B)
    const int prow = Lmax > 1 ? Lmax - 1 : 1;                      // parked rows per group: the finite layers
    double *ca = reinterpret_cast<double *>(smem);                 // [MPW*J][prow][CA_STRIDE]
    double *xs = ca + (size_t)MPW * J * prow * CA_STRIDE;          // [11][MPW]
    double *ys = xs + NEV_MAX * MPW;
    double *per = ys + NEV_MAX * MPW;                              // [K]
    float *mdl = reinterpret_cast<float *>(per + ((K + 1) & ~1));  // [4][Lmax][MPW]
    unsigned char *after = reinterpret_cast<unsigned char *>(mdl) + (((size_t)4 * Lmax * MPW * sizeof(float) + 15) & ~(size_t)15);
    double *cpl = reinterpret_cast<double *>(after); // [2][Kmax][MPW], only if a target has mode > 1

    for (int k = lane; k < K; k += BH_WAVE) per[k] = T.periods[k];
    // stage the models of this wave: consecutive lanes -> consecutive models (coalesced for
    // layer-major input), binary32 rounding like the f2py boundary
    for (int idx = lane; idx < Lmax * MPW; idx += BH_WAVE) {
        const int l = idx / MPW, mg = idx % MPW;
        const int sb = lo + wid * MPW + mg;
        const int b = sb < hi ? (perm ? perm[sb] : sb) : 0;
        float fd = 0.f, fa = 1.f, fb = 1.f, fr = 1.f;
        if (sb < hi && l < A.nlay[b]) {
            const ptrdiff_t o = (ptrdiff_t)b * T.sb + (ptrdiff_t)l * T.sl;
            fd = (float)T.h[o];
            fa = (float)T.vp[o];
            fb = (float)T.vs[o];
            fr = (float)T.rho[o];
        }
        mdl[(0 * Lmax + l) * MPW + mg] = fd;
        mdl[(1 * Lmax + l) * MPW + mg] = fa;
        mdl[(2 * Lmax + l) * MPW + mg] = fb;
        mdl[(3 * Lmax + l) * MPW + mg] = fr;
    }
    const int mmax = valid ? A.nlay[ib] : 2;
    int mtop = mmax;
    for (int off = 32; off > 0; off >>= 1) mtop = max(mtop, __shfl_xor(mtop, off));
    mtop = __builtin_amdgcn_readfirstlane(mtop);
    wave_sync();
    ModelLdsRt md;
    md.S = MPW;
    md.d = mdl + 0 * Lmax * MPW + g;
    md.a = mdl + 1 * Lmax * MPW + g;
    md.b = mdl + 2 * Lmax * MPW + g;
    md.rho = mdl + 3 * Lmax * MPW + g;
    const int llw = (md.Bf(0) <= 0.0f) ? 2 : 1;
    double *cam = ca + (size_t)slot * prow * CA_STRIDE; // this group's parked layers
    const bool par5 = (G >= 5) && !(Gflags & 0x100);
    const int gbase = slot * G; // first lane of this group
    // one layer count for the whole wavefront and no water layer: the recursion needs no masking
    const bool ragged = __ballot(mmax != mtop || llw != 1) != 0ull;
    const int col = li % 5;                          // the 5-vector component this lane owns

    constexpr bool FAST = FASTM != 0;
    constexpr bool BULK = FAST; // runs of plain bracket steps consumed in one go (the short-refinement build only)
    SearchT<0, NEV_MAX, FASTM, SIMPLE> S;
    S.XS = MPW;
    // RESTART (one model per wavefront, the build with both sequences, SwdMultiArgs::restart): when the guard of the short
    // refinement fires, the model starts again right here with the reference's sequence -- what the engine's re-run launch
    // would do for it afterwards, without the second launch (the guarded models of a sampler's window are its shallow ones:
    // their second search ends before the window's deep models do).  Same rows, same flags.
    bool refseq_now = T.refseq != 0, restarted = false;
    unsigned evals_before = 0u; // (evaluations of the abandoned first search: they count, as the re-run launch's would)
restart_with_the_reference_sequence:
    S.init(md, mmax, valid, T.igr, K, per, xs + g, ys + g, T.vel + (size_t)ib * T.ldv, li == 0 && rr == 0 && !spare,
           T.mode, cpl + g, cpl + (size_t)K * MPW + g, ifunc, CNTB && A.counted != 0, refseq_now, FA);
    S.evals += evals_before;

    // per-period constants of this lane's first layer (m = li) and of the half-space: they depend
    // on omega and the model only, not on the trial phase velocity -> recomputed when omega changes
    // (the build with both sequences AND the counted scan has no registers to keep them in: formed anew every round there)
    constexpr bool CACHE = !(CNTB && FASTM == 1);
    double c_omega = -1.0, c_xka = 0.0, c_xkb = 0.0, c_gammk = 0.0, h_xka = 0.0, h_xkb = 0.0, h_gammk = 0.0;
    double c_inv = 0.0; // fast arithmetic: 1 / rho (Rayleigh) resp. 1 / xmu (Love) of this lane's first layer
    const bool prof = PROF && (A.neval != nullptr);
    // (the phase clocks are a development aid; the one-model-per-wavefront build with both sequences and the counted scan has
    //  no registers for them)
    long long tA = 0, tB = 0, tS = 0, t0 = 0, t1c = 0, t2c = 0;
    constexpr bool CLK = PROF && !(ADAPT && CNTB && FASTM == 1);
    const bool clocks = CLK && (A.neval != nullptr);
    const unsigned long long w_start = clocks ? wall_clock64() : 0ull, c_start = clocks ? clock64() : 0ull;
    unsigned int nrounds = 0;
    // Two wavefronts share a SIMD; at equal priority the hardware serves the OLDER one first (MI355X_MICROARCH.md,
    // "two waves per SIMD"): the older runs its rounds in ~13 k cycles, the younger in ~17 k, finishes 30 % later and
    // sets the kernel's time.  The two therefore (a) alternate between a high and a low issue priority every
    // PRIO_BLOCK rounds, the two hardware wave slots of a SIMD in opposite phase -- each is the favoured one half of
    // the time -- and (b) tell each other how far they are: every PRIO_BLOCK rounds a wavefront publishes the
    // fraction of its period list its slowest model has behind it in a small board in global memory, indexed by
    // the physical SIMD, and reads its neighbour's; whoever is behind takes the high priority until they are level,
    // so that both reach the end of their lists together.  (Scheduling only: results do not depend on it.)
    unsigned hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const unsigned hw_slot = hw_id & 1u;
    const unsigned simd_key = ((xcc_id & 7u) << 12) | ((hw_id >> 4) & 0xfffu);
    unsigned *board = A.board ? A.board + 2u * simd_key : nullptr;
    const bool fair = (Gflags & 0x800) == 0;
    const unsigned total_p = (unsigned)(K * (T.mode > 0 ? T.mode : 1));
    while (__ballot(S.active) != 0ull) {
        if (fair && (nrounds & (PRIO_BLOCK - 1)) == 0) {
            bool high = (((nrounds / PRIO_BLOCK) ^ hw_slot) & 1u) != 0u;
            if (board != nullptr && !(Gflags & 0x400)) {
                // periods behind the slowest model of this wavefront, as a fraction (16 bits) of its list
                unsigned done = valid ? (S.active ? (unsigned)((S.iq - 1) * K + S.k) : total_p) : total_p;
                for (int off = 32; off > 0; off >>= 1) done = min(done, (unsigned)__shfl_xor((int)done, off));
                const unsigned frac = __builtin_amdgcn_readfirstlane(total_p ? (done << 12) / total_p : 4096u);
                const unsigned mine = (A.stamp << 16) | frac;
                // (plain store: written through the CU's L1 and KEPT in the XCD's L2 -- an agent-scope atomic store drops
                // the line, and every look at the board then went to memory: 25 MB per launch; the load bypasses L1.
                // Both wavefronts sit on the same CU, hence behind the same L2.)
                *reinterpret_cast<volatile unsigned *>(board + hw_slot) = mine;
                const unsigned other = __hip_atomic_load(board + (hw_slot ^ 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((other >> 16) == A.stamp && (other & 0xffffu) != frac) high = frac < (other & 0xffffu);
            }
            if (high) __builtin_amdgcn_s_setprio(3);
            else if (A.prio_low == 0) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(1);
        }
        ++nrounds;
        // All lanes take part in the evaluation (finished models compute on stale values).
        if (clocks) t0 = clock64();
        const double omg = S.omega;
        const int cb = rr * JL + (li % JL); // the trial this lane carries through the recursion
        // (Rayleigh wavefronts take the instantiations without the counted scan -- `ifunc` is uniform per wavefront)
        const double cev = (J * JL == 1) ? S.ceval : ((!CNTB || ifunc == 2) ? S.template candidate<false>(cb) : S.template candidate<CNTB>(cb));
        const double wvno = omg / cev;
        double del;
        int nv = -1; // Love: the packed mode count of this lane's trial (LoveCount)
        if (ifunc == 2) {
            double omega = omg;
            if (omega < 1.0e-4) omega = 1.0e-4;
            const double wvno2 = wvno * wvno;
            if (!CACHE || omega != c_omega) {
                c_omega = omega;
                if (li <= mmax - 2) {
                    const double am = md.A(li), bm = md.Bv(li);
                    c_xka = omega / am;
                    c_xkb = omega / bm;
                    const double t = bm / omega;
                    c_gammk = 2.0 * t * t;
                    if (FA) c_inv = fa::rcp(md.R(li));
                }
                const double ah = md.A(mmax - 1), bh = md.Bv(mmax - 1);
                h_xka = omega / ah;
                h_xkb = omega / bh;
                const double t = bh / omega;
                h_gammk = 2.0 * t * t;
            }
            // ---- phase A: layer terms, one layer per lane (strided by G) -----------------------
            for (int m = li; m <= mmax - 2; m += G) {
                if (m >= llw - 1) {
                    double xka, xkb, gammk;
                    if (m == li) {
                        xka = c_xka;
                        xkb = c_xkb;
                        gammk = c_gammk;
                    } else { // deep models: further rounds are computed on the fly
                        const double am = md.A(m), bm = md.Bv(m);
                        xka = omega / am;
                        xkb = omega / bm;
                        const double t = bm / omega;
                        gammk = 2.0 * t * t;
                    }
                    const double gam = gammk * wvno2;
                    const double dpth = md.D(m);
                    const double rho1 = md.R(m);
                    LayerTerms v;
                    Ca19 c;
                    if (FA) {
                        fa::layer_products(wvno, xka, xkb, dpth, v);
                        fa::ca19(c, wvno2, gam, gammk, rho1, (m == li) ? c_inv : fa::rcp(rho1), v);
                    } else {
                        double wvnop = wvno + xka;
                        double wvnom = fabs(wvno - xka);
                        const double ra = sqrt(wvnop * wvnom);
                        wvnop = wvno + xkb;
                        wvnom = fabs(wvno - xkb);
                        const double rb = sqrt(wvnop * wvnom);
                        layer_products(ra * dpth, rb * dpth, ra, rb, wvno, xka, xkb, dpth, v, LT);
                        rayleigh_ca19(c, wvno2, gam, gammk, rho1, v);
                    }
                    park_ca25(cam + (size_t)m * CA_STRIDE, c);
                }
            }
            // half-space E vector (surfdisp96.f:800-808), redundantly in every lane
            double e[5];
            if (FA) {
                fa::rayleigh_halfspace(e, wvno, wvno2, h_xka, h_xkb, h_gammk, md.R(mmax - 1));
            } else {
                const double xka = h_xka, xkb = h_xkb, gammk = h_gammk;
                double wvnop = wvno + xka;
                double wvnom = fabs(wvno - xka);
                const double ra = sqrt(wvnop * wvnom);
                wvnop = wvno + xkb;
                wvnom = fabs(wvno - xkb);
                const double rb = sqrt(wvnop * wvnom);
                const double gam = gammk * wvno2;
                const double gamm1 = gam - 1.0;
                const double rho1 = md.R(mmax - 1);
                e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
                e[1] = -rho1 * ra;
                e[2] = rho1 * (gamm1 - gammk * ra * rb);
                e[3] = rho1 * rb;
                e[4] = wvno2 - ra * rb;
            }
            wave_sync();
            if (clocks) t1c = clock64();
            // ---- phase B: the sequential recursion, bottom-up over the parked layers -----------
            {
                double e0[5] = {e[0], e[1], e[2], e[3], e[4]};
                DivRange dr;
                dr.reset();
                if (par5) {
                    if (ragged) rayleigh_chain_group<true, true, false, FA>(e, cam, col, gbase, mtop, mmax, llw, dr);
                    else rayleigh_chain_group<true, false, false, FA>(e, cam, col, gbase, mtop, mmax, llw, dr);
                } else {
                    if (ragged) rayleigh_chain_group<false, true, false, FA>(e, cam, col, gbase, mtop, mmax, llw, dr);
                    else rayleigh_chain_group<false, false, false, FA>(e, cam, col, gbase, mtop, mmax, llw, dr);
                }
                if (!FA && !dr.ok() && S.active) { // out-of-range operand somewhere: verbatim re-run (whole groups agree)
                    e[0] = e0[0]; e[1] = e0[1]; e[2] = e0[2]; e[3] = e0[3]; e[4] = e0[4];
                    rayleigh_chain_group<false, true, true>(e, cam, col, gbase, mtop, mmax, llw, dr);
                }
            }
            del = e[0];
            if (llw != 1) { // water layer on top (surfdisp96.f:850-866)
                const double xka = omega / md.A(0);
                const double wvnop = wvno + xka;
                const double wvnom = fabs(wvno - xka);
                const double ra = sqrt(wvnop * wvnom);
                const double dpth = md.D(0);
                const double rho1 = md.R(0);
                const double znul = 1.0e-5;
                LayerTerms v;
                if (FA) fa::layer_products(wvno, xka, wvno + 1.0, dpth, v); // (only w and cosp of the P wave are used)
                else layer_products(ra * dpth, znul, ra, znul, wvno, xka, znul, dpth, v, LT);
                const double w0 = -rho1 * v.w;
                del = v.cosp * e[0] + w0 * e[1];
            }
            wave_sync();
        } else {
            const double omega = omg;
            if (!CACHE || omega != c_omega) {
                c_omega = omega;
                if (li <= mmax - 2) {
                    c_xkb = omega / md.Bv(li);
                    if (FA) c_inv = fa::rcp(md.R(li) * md.Bv(li) * md.Bv(li));
                }
                const double beta1 = md.Bv(mmax - 1);
                h_xkb = omega / beta1;
                h_gammk = 1.0 / (beta1 * beta1); // e2 of the half-space (surfdisp96.f:731)
            }
            // ---- phase A (Love): cosq, y, z, xmu per layer, for each of the group's JL trials ----------
            for (int jj = 0; jj < JL; ++jj) {
                const double wv = (JL == 1) ? wvno : omg / S.template candidate<CNTB>(rr * JL + jj);
                for (int m = li; m <= mmax - 2; m += G) {
                    if (m >= llw - 1) {
                        const double beta1 = md.Bv(m);
                        const double rho1 = md.R(m);
                        const double dm = md.D(m);
                        const double xmu = rho1 * beta1 * beta1;
                        const double xkb = (m == li) ? c_xkb : omega / beta1;
                        double cosq, y, z, fl = 0.0;
                        if (FA) {
                            double q;
                            fa::love_terms(wv, xkb, dm, cosq, y, z, q);
                            if (CNTB) fl = (wv < xkb) ? love_zero_floor(q) : 0.0;
                            double2 *dst = reinterpret_cast<double2 *>(cam + (size_t)m * CA_STRIDE + LOVE_TERMS * jj);
                            dst[0] = make_double2(cosq, y * ((m == li) ? c_inv : fa::rcp(xmu))); // (the products of the recursion's
                            dst[1] = make_double2(z * xmu, 0.0);                                  //  step are formed here, in parallel)
                            dst[2] = make_double2(0.0, fl);
                            continue;
                        }
                        const double wvnop = wv + xkb;
                        const double wvnom = fabs(wv - xkb);
                        const double rb = sqrt(wvnop * wvnom);
                        const double q = dm * rb;
                        if (wv < xkb) {
                            double sinq;
                            bh_sincos(q, &sinq, &cosq, LT);
                            y = sinq / rb;
                            z = -rb * sinq;
                            if (CNTB) fl = love_zero_floor(q);
                        } else if (wv == xkb) {
                            cosq = 1.0;
                            y = dm;
                            z = 0.0;
                        } else {
                            double fac = 0.0;
                            if (q < 16.0) fac = bh_exp(-2.0 * q, LT);
                            cosq = (1.0 + fac) * 0.5;
                            const double sinq = (1.0 - fac) * 0.5;
                            y = sinq / rb;
                            z = rb * sinq;
                        }
                        double2 *dst = reinterpret_cast<double2 *>(cam + (size_t)m * CA_STRIDE + LOVE_TERMS * jj);
                        dst[0] = make_double2(cosq, y);
                        dst[1] = make_double2(z, xmu);
                        dst[2] = make_double2(bh_rcp_refined(xmu), fl);
                    }
                }
            }
            double e1, e2;
            LoveCount lc;
            {
                const double rho1 = md.R(mmax - 1);
                const double xkb = h_xkb;
                const double wvnop = wvno + xkb;
                const double wvnom = fabs(wvno - xkb);
                double rb;
                if (FA) {
                    double t_;
                    const double r2 = wvnop * wvnom;
                    fa::sqrt_rsqrt(r2 > 1.0e-290 ? r2 : 1.0, rb, t_);
                    rb = r2 > 1.0e-290 ? rb : 0.0;
                } else {
                    rb = sqrt(wvnop * wvnom);
                }
                e1 = rho1 * rb;
                e2 = h_gammk;
                lc.reset(wvno > xkb);
            }
            wave_sync();
            if (clocks) t1c = clock64();
            {
                const double s1 = e1, s2 = e2;
                const LoveCount lc0 = lc;
                DivRange dr;
                dr.reset();
                const double *camt = cam + LOVE_TERMS * (li % JL); // this lane's trial
                if (ragged) love_chain_group<true, false, CNTB, FA>(e1, e2, camt, mtop, mmax, llw, dr, lc);
                else love_chain_group<false, false, CNTB, FA>(e1, e2, camt, mtop, mmax, llw, dr, lc);
                if (!FA && !dr.ok() && S.active) {
                    e1 = s1;
                    e2 = s2;
                    lc = lc0;
                    love_chain_group<true, true, CNTB>(e1, e2, camt, mtop, mmax, llw, dr, lc);
                }
            }
            del = e1;
            if (CNTB) nv = lc.packed(e1, e2);
            wave_sync();
        }
        if (clocks) t2c = clock64();
        auto consume = [&](auto cnt_tag) {
            constexpr bool CNT = decltype(cnt_tag)::value;
        // Every lane of the model can read all J (velocity, value) pairs; the search consumes them for
        // as long as its next request is the very velocity (at the same omega) the next group evaluated.
        if (BULK && !CNT && J * JL > 2) { // (measured: pays from three trials per round on; with two it costs 6 % at B = 4096; Love's
                                          //  counted scan hardly ever takes plain steps: not compiled in for its wavefronts)
            // The same consumption with the runs of plain bracket steps taken in one go.  A trial is PLAIN for its model
            // when the search is scanning (ST_STEP), the value has the sign of del1, the velocity is inside the scan's
            // bounds and the request after it is again the plain next step: consuming it only moves (c1, del1) on.
            // Every lane judges the trial it carried; a ballot hands every model the verdicts of all its trials; the
            // leading run of plain trials is consumed by copying the last one's (velocity, value) -- what advance()
            // would have left after stepping through them one by one.  Anything else goes through advance().
            const int Jtot = J * JL;
            bool live = S.active;
            int jn = 0; // this model's next trial
            while (__ballot(live) != 0ull) {
                // trial jn must be the very request (trial 0 IS the pending request)
                const int tj = live ? jn : 0;
                const int srcn = (g * J + tj / JL) * G + (tj % JL);
                const double cj = __shfl(cev, srcn), dj = __shfl(del, srcn);
                const int nj = CNT ? __shfl(nv, srcn) : -1;
                if (jn > 0) live = live && S.ceval == cj && S.omega == omg;
                const double nxt = (S.idir > 0) ? cev + S.dc : cev - S.dc;
                // (a step that reaches a half-space velocity goes through advance(): the guard's probes)
                const bool plain = S.active && S.st == ST_STEP && !signs_differ(S.del1, del) && (!FA || fabs(del) >= fa::SIGN_FLOOR) &&
                                   !(cev < S.cm || cev >= S.betmxd + S.dc) && nxt > S.clow && fmax(cev, S.c1) < S.vsafe;
                const unsigned long long pm = __ballot(plain);
                bool run = false;
                if (live && S.st == ST_STEP) {
                    int t = jn;
                    while (t < Jtot && ((pm >> ((g * J + t / JL) * G + (t % JL))) & 1ull)) ++t;
                    run = t > jn;
                    if (run) {
                        const int src = (g * J + (t - 1) / JL) * G + ((t - 1) % JL);
                        const double cl = __shfl(cev, src), dl = __shfl(del, src);
                        // (the point the run leaves behind: the trial before the last, or -- a run of one -- the old c1)
                        const int srcp = (t - jn >= 2) ? (g * J + (t - 2) / JL) * G + ((t - 2) % JL) : src;
                        const double cq = __shfl(cev, srcp), dq = __shfl(del, srcp);
                        S.cp = (t - jn >= 2) ? cq : S.c1;
                        S.delp = (t - jn >= 2) ? dq : S.del1;
                        S.have_p = true;
                        S.c1 = cl;
                        S.del1 = dl;
                        S.del2 = dl;
                        S.c2 = (S.idir > 0) ? cl + S.dc : cl - S.dc;
                        S.ceval = S.c2;
                        S.evals += (unsigned)(t - jn);
                        S.isteps += t - jn;
                        jn = t;
                    }
                }
                if (live && !run) {
                    S.template advance<CNT>(dj, nj);
                    ++jn;
                }
                live = live && S.active && jn < Jtot;
            }
        } else if (CNT && J * JL > 2) { // (with two trials per round the order is the only one possible)
            // The counted scan asks for grid points in an order that depends on the values (start value, jump target, then an
            // index search): every trial lane carries one of the likely ones (SearchT::candidate), and a trial's value is taken
            // whenever the search asks for exactly its velocity -- in whatever order, each trial at most once.  Trial 0 IS the
            // pending request.  (Which values the search consumes does not depend on it: the requests are the search's own.)
            bool live = S.active;
            unsigned long long used = 0ull;
            int src = (g * J) * G; // a lane that carried trial 0
            const bool rep = !spare && li < JL; // one lane per trial answers
            const unsigned long long mine = (LPM >= 64) ? ~0ull : (((1ull << LPM) - 1ull) << (g * LPM)); // this model's lanes
            for (int it = 0; it < J * JL; ++it) {
                if (__ballot(live) == 0ull) break;
                const double dj = __shfl(del, src);
                const int nj = __shfl(nv, src);
                if (live) {
                    used |= 1ull << src;
                    S.template advance<CNT>(dj, nj);
                }
                live = live && S.active && S.omega == omg;
                const unsigned long long m = __ballot(live && rep && cev == S.ceval) & mine & ~used;
                live = live && m != 0ull;
                src = live ? (int)__builtin_ctzll(m) : src;
            }
        } else {
            bool live = S.active;
            const int Jtot = J * JL;
            for (int j = 0; j < Jtot; ++j) {
                double dj = del;
                int nj = nv;
                if (Jtot > 1) {
                    const int src = (g * J + j / JL) * G + (j % JL); // a lane that carried trial j
                    const double cj = __shfl(cev, src);
                    dj = __shfl(del, src);
                    if (CNT) nj = __shfl(nv, src);
                    // trial 0 IS the pending request (consumed unconditionally, also when a broken model
                    // has driven the search to NaN); a later trial only if the search now asks for it
                    if (j > 0) live = live && S.active && S.ceval == cj && S.omega == omg;
                }
                if (__ballot(live) == 0ull) break;
                if (FAST) {
                    // a plain bracket step (three in four of this build's transitions): the value has del1's sign, the
                    // velocity is inside the scan's bounds and the next request is the next grid point -- what advance()
                    // does then, without its way through the continuation tags
                    const double nx = (S.idir > 0) ? S.c2 + S.dc : S.c2 - S.dc;
                    const bool plain = live && S.st == ST_STEP && !signs_differ(S.del1, dj) && (!FA || fabs(dj) >= fa::SIGN_FLOOR) &&
                                       !(S.c2 < S.cm || S.c2 >= S.betmxd + S.dc) && nx > S.clow && fmax(S.c1, S.c2) < S.vsafe;
                    if (plain) {
                        S.cp = S.c1;      // (the point the scan leaves behind, see SearchT::step_done)
                        S.delp = S.del1;
                        S.have_p = true;
                        S.del2 = dj;
                        S.c1 = S.c2;
                        S.del1 = dj;
                        S.c2 = nx;
                        S.ceval = nx;
                        ++S.evals;
                        ++S.isteps;
                    } else if (live) {
                        S.template advance<CNT>(dj, nj);
                    }
                } else if (live) S.template advance<CNT>(dj, nj);
            }
        }
        };
        if (!CNTB || ifunc == 2) consume(std::false_type{});
        else consume(std::integral_constant<bool, CNTB>{});
        if (clocks) {
            const long long t3 = clock64();
            tA += t1c - t0;
            tB += t2c - t1c;
            tS += t3 - t2c;
        }
    }
    if (ADAPT && FASTM == 1 && A.restart != 0 && !restarted && __ballot(valid && S.has(S.F_GUARD)) != 0ull) {
        restarted = true;
        refseq_now = true;
        evals_before = S.evals;
        if (valid && li == 0 && rr == 0 && !spare && T.gcount != nullptr) {
            atomicAdd(T.gcount + BH_MAX_TARGETS, 1);     // this call: restarted in place
            atomicAdd(T.gcount + 2 * BH_MAX_TARGETS, 1); // cumulative
        }
        goto restart_with_the_reference_sequence;
    }
    if (board != nullptr) *reinterpret_cast<volatile unsigned *>(board + hw_slot) = (A.stamp << 16) | 0xffffu;
    if (valid && li == 0 && rr == 0 && !spare) {
        T.err[ib] = S.errflag;
        if (!SIMPLE && T.igr == 2 && T.first != nullptr) T.first[ib] = S.del1st; // (the chain of a group velocity's first roots: SwdKernelArgs)
        if (FAST && S.has(S.F_GUARD) && T.gcount != nullptr) { // to be run again with the reference's sequence
            T.glist[atomicAdd(T.gcount, 1)] = ib;
            atomicAdd(T.gcount + 2 * BH_MAX_TARGETS, 1); // (cumulative, for bh_engine_guard_stats)
        }
    }
    if (prof) {
        unsigned long long tot = (li == 0 && rr == 0 && !spare) ? S.evals : 0u;
        unsigned long long lps = tot * (unsigned long long)(valid ? mmax - 1 : 0);
        for (int off = 32; off > 0; off >>= 1) {
            tot += __shfl_xor(tot, off);
            lps += __shfl_xor(lps, off);
        }
        if (lane == 0) {
            // [0] secular evaluations; per wave type ([8] Rayleigh / [9] Love) evaluations and ([10] / [11]) layer-
            // propagator steps = evaluations x finite layers of the model (the flop model of SURVEY.md 8(d))
            atomicAdd(A.neval, tot);
            atomicAdd(A.neval + (ifunc == 2 ? 8 : 9), tot);
            atomicAdd(A.neval + (ifunc == 2 ? 10 : 11), lps);
            // development aid: wave-cycles per phase, [1..3] Rayleigh A/B/state, [4..6] Love
            const int o = (ifunc == 2) ? 1 : 4;
            if (clocks) {
                atomicAdd(A.neval + o, (unsigned long long)tA);
                atomicAdd(A.neval + o + 1, (unsigned long long)tB);
                atomicAdd(A.neval + o + 2, (unsigned long long)tS);
            }
            const unsigned long long widx = atomicAdd(A.neval + 7, 1ull);
            if (clocks && widx < BH_TRACE_WAVES) { // development aid: one record per wavefront (tools/gpu_trace.py)
                unsigned long long *r = A.neval + BH_COUNTER_WORDS + 4 * widx;
                unsigned hwid, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                hwid = (hwid & 0xffffu) | ((xcc & 0xfu) << 16);
                r[0] = w_start;
                r[1] = wall_clock64();
                r[2] = ((unsigned long long)(clock64() - c_start) & 0xffffffffffull) | ((unsigned long long)(unsigned)(blockIdx.x * WPB + wave) << 40); // cycles | grid wavefront index << 40
                r[3] = (unsigned long long)nrounds | ((unsigned long long)ifunc << 32) | ((unsigned long long)hwid << 36);
            }
        }
    }
    if (PASSES && A.rerun != 0) { // the next entries of the list
        ++pass;
        goto next_pass;
    }
}

#if !defined(BH_GROUP_FA_TU) && !defined(BH_GROUP_BIG_TU) && !defined(BH_GROUP_ADAPT_TU)
size_t group_lds_bytes(int G, int J, int Lmax, int Kmax, int maxmode)
{
    const int MPW = BH_WAVE / (G * J);
    return ((size_t)MPW * J * (Lmax > 1 ? Lmax - 1 : 1) * CA_STRIDE + (size_t)2 * NEV_MAX * MPW +
            (size_t)((Kmax + 1) & ~1)) * sizeof(double) +
           (((size_t)4 * Lmax * MPW * sizeof(float) + 15) & ~(size_t)15) +
           (maxmode > 1 ? (size_t)2 * Kmax * MPW * sizeof(double) : 0);
}
#endif

} // namespace

#ifdef BH_GROUP_BIG_TU
// The build that needs more than 256 registers (this translation unit: swd_group_big.hip, ONE wavefront per SIMD as its register
// budget): one model per wavefront, both sequences, the counted Love scan AND the counters and clocks -- an instrumented launch of a
// sampler's window under BH_SEARCH_FAST_RAYLEIGH; counters and clocks are what it is for, not speed.
void bh_launch_swd_group_big(const SwdMultiArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t stream, int redundant, int wave_lds)
{
    hipLaunchKernelGGL((swd_group_kernel<GROUP_WPB, 1, true, true, true, true>), grid, block, lds, stream, a, redundant, wave_lds);
}
#elif defined(BH_GROUP_ADAPT_TU)
// The one-model-per-wavefront builds (ADAPT: a sampler's windows, single models, the re-run of guarded models) in a translation unit
// of their own (swd_group_adapt.hip: the same source, the same flags) -- a third of this file's instantiations: it halves the
// build's longest compile.  fm = the sequences compiled in (FASTM), pr = with counters and clocks, cn = with the counted Love scan.
void bh_launch_swd_group_adapt(const SwdMultiArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t stream, int redundant, int wave_lds,
                               int fm, bool pr, bool cn)
{
#define BH_AD_(FM, PR, CN) hipLaunchKernelGGL((swd_group_kernel<GROUP_WPB, FM, true, PR, true, CN>), grid, block, lds, stream, a, redundant, wave_lds)
#define BH_AD(FM, PR) do { if (cn) BH_AD_(FM, PR, true); else BH_AD_(FM, PR, false); } while (0)
    if (fm == 2) {
        if (pr) BH_AD(2, true);
        else BH_AD(2, false);
    } else if (fm == 1) {
        if (pr) BH_AD_(1, true, false); // (with the counted scan as well: swd_group_big.hip)
        else BH_AD(1, false);
    } else {
        if (pr) BH_AD(0, true);
        else BH_AD(0, false);
    }
#undef BH_AD
#undef BH_AD_
}
#elif defined(BH_GROUP_FA_TU)
// The launches of the builds with the fast arithmetic (this translation unit: swd_group_fa.hip).
void bh_launch_swd_group_fa(const SwdMultiArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t stream, int redundant, int wave_lds,
                            bool adapt, bool counted, bool cntb)
{
#define BH_FA_(PR, AD, CN) hipLaunchKernelGGL((swd_group_kernel<GROUP_WPB, 2, true, PR, AD, CN, true>), grid, block, lds, stream, a, redundant, wave_lds)
#define BH_FA(PR, AD) do { if (cntb) BH_FA_(PR, AD, true); else BH_FA_(PR, AD, false); } while (0)
    if (adapt) {
        if (counted) BH_FA(true, true);
        else BH_FA(false, true);
    } else {
        if (counted) BH_FA(true, false);
        else BH_FA(false, false);
    }
#undef BH_FA
#undef BH_FA_
}
#else

// LDS of one workgroup = shared libm tables + GROUP_WPB wavefront regions
size_t bh_swd_group_lds_bytes(int G, int J, int Lmax, int Kmax, int maxmode)
{
    return LIBM_TAB_PAD + GROUP_WPB * ((group_lds_bytes(G, J, Lmax, Kmax, maxmode) + 15) & ~(size_t)15);
}

// A wavefront's LDS region small enough for 8 wavefronts (4 workgroups) per CU of 160 KB
constexpr size_t WAVE_LDS_TARGET = (160 * 1024 / 4 - LIBM_TAB_PAD) / GROUP_WPB;
constexpr size_t WG_LDS_CAP = 64 * 1024;

namespace {
// Load ranks of the wavefronts of a launch for the SIMD-pairing order (SwdPairWork).  Placement rule of the workgroup
// dispatcher, read off the wavefront trace (tools/gpu_trace.py, MAP=...; identical from run to run): workgroup g of a
// one-dimensional grid runs on CU g mod ncu in pass g / ncu, and its two wavefronts sit on SIMD (pass + j) mod 4 -- so SIMD
// s of a CU holds wavefront 0 of its pass-s workgroup and wavefront 1 of its pass-(s-1) workgroup, and the SIMDs whose
// partner workgroup does not exist (the last pass is not full) hold one wavefront only.
// Every wavefront gets a wanted load quantile q (0 = longest models): 0 for a wavefront alone on its SIMD, u and 1 - u for
// the two of a pair (u spread over (0, 1/2) across the pairs); per target the wavefronts are ranked by q.
bool build_slot_ranks(int n0, int n1, int ncu, std::vector<int32_t> rank[2])
{
    const int NW = n0 + n1, nwg = (NW + 1) / 2;
    if (ncu <= 0 || nwg > 4 * ncu) return false; // more than one round of workgroups: placement not fixed by the index
    auto target_of = [&](int W, int &ty, int &wid) {
        if (n1 > 0) {
            const long long N = NW;
            const int l0 = (int)(((long long)W * n1) / N), l1 = (int)((((long long)W + 1) * n1) / N);
            ty = (l1 > l0) ? 1 : 0;
            wid = (l1 > l0) ? l0 : W - l0;
        } else {
            ty = 0;
            wid = W;
        }
    };
    std::vector<int> on_simd[2]; // per (cu, simd): wavefront 0 / wavefront 1 member (grid wavefront index or -1)
    on_simd[0].assign((size_t)ncu * 4, -1);
    on_simd[1].assign((size_t)ncu * 4, -1);
    for (int W = 0; W < NW; ++W) {
        const int g = W / 2, j = W % 2, cu = g % ncu, pass = g / ncu, simd = (pass + j) % 4;
        on_simd[j][(size_t)cu * 4 + simd] = W;
    }
    std::vector<double> q((size_t)NW, 0.0);
    int npairs = 0;
    for (size_t k = 0; k < on_simd[0].size(); ++k) npairs += (on_simd[0][k] >= 0 && on_simd[1][k] >= 0);
    int ip = 0;
    for (size_t k = 0; k < on_simd[0].size(); ++k) {
        const int a = on_simd[0][k], b = on_simd[1][k];
        if (a >= 0 && b >= 0) {
            const double u = (ip + 0.5) / (2.0 * npairs);
            q[(size_t)((ip & 1) ? a : b)] = u;        // (alternating which member takes the long share spreads the long
            q[(size_t)((ip & 1) ? b : a)] = 1.0 - u;  //  models evenly over the two targets)
            ++ip;
        } else if (a >= 0) {
            q[(size_t)a] = 0.0;
        } else if (b >= 0) {
            q[(size_t)b] = 0.0;
        }
    }
    const int nw[2] = {n0, n1};
    for (int t = 0; t < 2; ++t) {
        std::vector<std::pair<double, int>> v;
        for (int W = 0; W < NW; ++W) {
            int ty, wid;
            target_of(W, ty, wid);
            if (ty == t && wid < nw[t] - 1) v.emplace_back(q[(size_t)W], wid); // (the last wavefront keeps the tail)
        }
        std::sort(v.begin(), v.end());
        rank[t].assign((size_t)(nw[t] > 0 ? nw[t] : 1), 0);
        for (size_t r = 0; r < v.size(); ++r) rank[t][(size_t)v[r].second] = (int32_t)r;
    }
    return true;
}
} // namespace

int bh_launch_swd_group(const SwdMultiArgs &a0, int G0, hipStream_t stream, SwdLaunchInfo *info, int wpb, SwdPairWork *pair)
{
    wpb = GROUP_WPB; // (four wavefronts per workgroup were the co-resident receiver-function experiment of rounds 3-5: gone)
    int kmax = 0, maxmode = 1;
    for (int t = 0; t < a0.ntargets; ++t) {
        kmax = a0.t[t].K > kmax ? a0.t[t].K : kmax;
        maxmode = a0.t[t].mode > maxmode ? a0.t[t].mode : maxmode;
    }
    const BhTuning &tun = bh_tuning(); // (experiment switches, bh_tuning.h)
    const int redundant = (tun.swd_redundant ? 0x100 : 0) | (tun.swd_no_board ? 0x400 : 0) | (tun.swd_no_fair ? 0x800 : 0);
    SwdMultiArgs a = a0;
    const bool two = a0.split != nullptr && a0.Lcut < a0.Lmax;
    if (!two) a.split = nullptr;
    size_t wave_lds = 0;
    int nwaves = 1;
    for (int cls = two ? 0 : 1; cls <= 1; ++cls) {
        const int rows = (two && cls == 1) ? a0.Lcut : a0.Lmax;
        // fewer models per wavefront (more lanes per model) until the parked layers fit: first the
        // residency target, at the latest the 64 KB a workgroup may ask for
        int G = G0;
        auto trials = [&](int g, int t) {
            int J = a.t[t].look > 1 ? a.t[t].look : 1;
            while (J > 1 && g * J > BH_WAVE) --J;
            return J;
        };
        auto wave_bytes = [&](int g) {
            size_t w = 0;
            for (int t = 0; t < a.ntargets; ++t) {
                const size_t l = (group_lds_bytes(g, trials(g, t), rows, kmax, maxmode) + 15) & ~(size_t)15;
                w = l > w ? l : w;
            }
            return w;
        };
        auto waves = [&](int g) {
            long w = 0;
            for (int t = 0; t < a.ntargets; ++t) {
                const int mpw = BH_WAVE / (g * trials(g, t));
                w += (a.B + mpw - 1) / mpw;
            }
            return w;
        };
        auto most_models = [&](int g) {
            int m = 1;
            for (int t = 0; t < a.ntargets; ++t) {
                const int mpw = BH_WAVE / (g * trials(g, t));
                m = mpw > m ? mpw : m;
            }
            return m;
        };
        // a batch that leaves half the chip idle anyway: one lane per layer of the deepest model of the
        // class (a single pass over the layers) instead of lanes for the typical depth
        // (only where that costs neither look-ahead nor residency)
        for (int Gwide = rows - 1 > 16 ? 16 : rows - 1; Gwide > G; --Gwide) {
            bool same = waves(Gwide) <= 256;
            for (int t = 0; t < a.ntargets; ++t) same = same && trials(Gwide, t) == trials(G, t);
            if (same) {
                G = Gwide;
                break;
            }
        }
        while (most_models(G) > 1 && wave_bytes(G) > WAVE_LDS_TARGET) G += 1;
        // (the cap is what a workgroup of GROUP_WPB wavefronts may ask for by default; the same models per wavefront with 4)
        while (G < BH_WAVE && LIBM_TAB_PAD + GROUP_WPB * wave_bytes(G) > WG_LDS_CAP) G += 1;
        if (LIBM_TAB_PAD + GROUP_WPB * wave_bytes(G) > WG_LDS_CAP) return -1;
        a.rows[cls] = rows;
        a.lanes[cls] = G;
        const size_t wb = wave_bytes(G);
        wave_lds = wb > wave_lds ? wb : wave_lds;
        for (int t = 0; t < a.ntargets; ++t) {
            const int mpw = BH_WAVE / (G * trials(G, t));
            const int nx = (a.B + mpw - 1) / mpw; // worst case: the whole batch is in this class
            nwaves = nx > nwaves ? nx : nwaves;
        }
    }
    if (!two) {
        a.rows[0] = a.rows[1];
        a.lanes[0] = a.lanes[1];
    }
    dim3 grid((nwaves + wpb - 1) / wpb, a.ntargets, two ? 2 : 1);
    a.wg_n0 = a.wg_n1 = 0;
    const bool no_mix = tun.swd_no_mix != 0;
    if (a.ntargets == 2 && !two && !no_mix && !a.rerun) { // two targets, one depth class: interleave their wavefronts (see the kernel)
        int n[2];
        for (int t = 0; t < 2; ++t) {
            int J = a.t[t].look > 1 ? a.t[t].look : 1;
            while (J > 1 && a.lanes[1] * J > BH_WAVE) --J;
            const int mpw = BH_WAVE / (a.lanes[1] * J);
            n[t] = (a.B + mpw - 1) / mpw; // wavefronts of this target
        }
        a.wg_n0 = n[0];
        a.wg_n1 = n[1];
        grid = dim3((n[0] + n[1] + wpb - 1) / wpb, 1, 1);
    }
    // SIMD-pairing order of the models (SwdPairWork): a one-dimensional grid of one class, workgroups of two wavefronts
    if (pair != nullptr && !two && wpb == 2 && grid.y == 1 && grid.z == 1 && a.ntargets <= 2 && bh_pair_order_fits(a.B)) {
        int n[2] = {0, 0}, mpw[2] = {1, 1};
        for (int t = 0; t < a.ntargets; ++t) {
            int J = a.t[t].look > 1 ? a.t[t].look : 1;
            while (J > 1 && a.lanes[1] * J > BH_WAVE) --J;
            mpw[t] = BH_WAVE / (a.lanes[1] * J);
            n[t] = (a.B + mpw[t] - 1) / mpw[t];
        }
        bool ok = true, geom = true;
        const bool by_depth_only = false; // (whether the pairing applies is the engine's decision: launch_swd_jobs)
        if (!by_depth_only && (pair->key_n0 != n[0] || pair->key_n1 != n[1] || pair->key_wpb != wpb)) { // (rare: the batch shape changed)
            std::vector<int32_t> rank[2];
            geom = build_slot_ranks(n[0], n[1], pair->ncu, rank); // false: placement unknown -> plain sorted order
            for (int t = 0; ok && geom && t < a.ntargets; ++t) {
                if (pair->cap_rank[t] < n[t]) {
                    if (pair->slot_rank[t]) (void)hipFree(pair->slot_rank[t]);
                    pair->slot_rank[t] = nullptr;
                    ok = hipMalloc((void **)&pair->slot_rank[t], (size_t)(n[t] + 64) * sizeof(int32_t)) == hipSuccess;
                    pair->cap_rank[t] = ok ? n[t] + 64 : 0;
                }
                if (ok) {
                    ok = hipStreamSynchronize(stream) == hipSuccess && // (an earlier launch may still read the old table)
                         hipMemcpy(pair->slot_rank[t], rank[t].data(), (size_t)n[t] * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
                }
            }
            pair->key_n0 = (ok && geom) ? n[0] : -1;
            pair->key_n1 = (ok && geom) ? n[1] : -1;
            pair->key_wpb = (ok && geom) ? wpb : -1;
        }
        if (ok && pair->cap_perm < a.B) {
            for (int t = 0; t < 2; ++t) {
                if (pair->perm[t]) (void)hipFree(pair->perm[t]);
                pair->perm[t] = nullptr;
                ok = ok && hipStreamSynchronize(stream) == hipSuccess &&
                     hipMalloc((void **)&pair->perm[t], (size_t)(a.B + a.B / 4 + 64) * sizeof(int32_t)) == hipSuccess;
            }
            pair->cap_perm = ok ? a.B + a.B / 4 + 64 : 0;
        }
        if (ok && by_depth_only) {
            bh_launch_order(a.B, a.nlay, pair->perm[0], a.Lmax, nullptr, stream);
            a.perm = pair->perm[0];
        } else if (ok) {
            PairOrderTarget tg[2];
            for (int t = 0; t < a.ntargets; ++t) tg[t] = PairOrderTarget{mpw[t], n[t], geom ? pair->slot_rank[t] : nullptr, pair->perm[t]};
            bh_launch_pair_order(a.B, a.Lmax, a.nlay, a.t[0].vs, a.t[0].sl, a.t[0].sb, a.ntargets, tg, stream);
            for (int t = 0; t < a.ntargets; ++t) a.t[t].perm = pair->perm[t];
        }
        (void)hipGetLastError();
    }
    // One model per wavefront for every target (a chain window, a single model), one class, the usual targets: every wavefront
    // sizes its lane groups and trials for its own model (ADAPT, see the kernel) within the region all wavefronts can be
    // resident with.
    const bool no_adapt = tun.swd_no_adapt != 0;
    // (capacity up to 32 rows: at least two trials fit the region; deeper arrays keep the launcher's own choice of fewer, wider
    // lane groups in a larger region)
    bool adapt = a.adapt_ok && !no_adapt && !two && wpb == GROUP_WPB && a.Lmax <= 32 && a.Lmax >= 2;
    {
        bool plain = true;
        for (int t = 0; t < a.ntargets; ++t) {
            int J = a.t[t].look > 1 ? a.t[t].look : 1;
            while (J > 1 && a.lanes[1] * J > BH_WAVE) --J;
            adapt = adapt && BH_WAVE / (a.lanes[1] * J) == 1;
            plain = plain && a.t[t].igr == 0 && a.t[t].mode <= 1 && a.t[t].K <= 64;
        }
        adapt = adapt && plain;
        if (adapt) wave_lds = WAVE_LDS_TARGET & ~(size_t)15;
    }
    const size_t lds = LIBM_TAB_PAD + wpb * wave_lds;
    if (info != nullptr) {
        info->workgroups = grid.x * grid.y * grid.z;
        info->waves = (a.wg_n1 > 0) ? (long)a.wg_n0 + a.wg_n1 : (long)nwaves * a.ntargets;
        info->lds = lds;
    }
    // build: 0 = reference sequence; with the short refinement asked for: 2 = no group-velocity target in the launch (nevill
    // not compiled in), 1 = mixed (group-velocity targets keep the reference sequence), 0 = group-velocity targets only.
    // simple: fundamental-mode phase velocities only (builds 0 and 2; no second root, no mode loop in the kernel).
    bool any_phase = false, any_group = false, any_modes = false, any_refseq = false;
    for (int t = 0; t < a.ntargets; ++t) {
        any_phase = any_phase || (a.t[t].igr == 0 && !a.t[t].refseq);
        any_group = any_group || a.t[t].igr != 0;
        any_refseq = any_refseq || (a.t[t].igr == 0 && a.t[t].refseq);
        any_modes = any_modes || a.t[t].mode > 1;
    }
    // (one model per wavefront and SwdMultiArgs::restart: the build with both sequences, guarded models restart in place)
    const bool no_restart = tun.swd_no_restart != 0;
    // (not with the fast arithmetic: the reference's sequence runs in the reference's arithmetic -- the re-run launch)
    const bool restart = adapt && a.restart != 0 && a.fast && any_phase && !no_restart && !(a.farith != 0 && !any_group && !any_refseq && !any_modes);
    a.restart = restart ? 1 : 0;
    if (info != nullptr) info->restarts_in_place = restart ? 1 : 0;
    const int build = (a.fast && any_phase) ? ((any_group || any_refseq || restart) ? 1 : 2) : 0;
    const bool no_simple = tun.swd_no_simple != 0;
    const bool simple = !any_group && !any_modes && !no_simple;
    a.fast = build;
    const dim3 block(BH_WAVE * wpb);
    const bool counted = a.neval != nullptr;
    // The builds with the counted scan (CNTB, see the kernel): asked for, a Love target in the launch, and the launch is of the
    // kind it pays in -- one model per wavefront, or Love targets only.  BH_SWD_SCAN_ALWAYS=1: wherever asked for (measurements).
    bool any_love = false, all_love = true;
    for (int t = 0; t < a.ntargets; ++t) {
        any_love = any_love || a.t[t].iwave == 1;
        all_love = all_love && a.t[t].iwave == 1;
    }
    // a.counted: 1 = wherever a Love target is (BH_SCAN_COUNTED), 2 = where it pays (BH_SCAN_AUTO): with several models per
    // wavefront, launches of Love targets only (B = 4096: 2.06 -> 1.71 ms) and Rayleigh + Love launches of the short refinement
    // (there the LOVE wavefronts set the time: 379 rounds against the Rayleigh wavefronts' 302; counted 149 rounds of twice the
    // length: c2 2.51 -> 2.30 ms, c3 2.60 -> 2.49).  One model per wavefront (the trial lanes already walk the scan seven
    // steps a round: 1.25 -> 1.27 ms per window of the chains) and Rayleigh + Love launches of the reference's sequence (the
    // Rayleigh wavefronts set the time: c2 3.37 -> 3.47 ms) do not gain.
    // (not in a launch of several models per wavefront that mixes both refinements: that build would spill; one model per
    //  wavefront -- the chains' windows -- has it)
    const bool cntb = any_love && wpb == GROUP_WPB && !(build == 1 && !adapt) && (a.counted == 1 || (a.counted == 2 && !adapt && (all_love ? build != 1 : build == 2)));
    a.counted = cntb ? 1 : 0;
#define BH_GROUP_LAUNCH_(WP, FM, SI, PR, AD, CN) hipLaunchKernelGGL((swd_group_kernel<WP, FM, SI, PR, AD, CN>), grid, block, lds, stream, a, redundant, (int)wave_lds)
#define BH_GROUP_LAUNCH(WP, FM, SI, PR) do { if (cntb) BH_GROUP_LAUNCH_(WP, FM, SI, PR, false, true); else BH_GROUP_LAUNCH_(WP, FM, SI, PR, false, false); } while (0)
#define BH_GROUP_LAUNCH_ADAPT(FM, PR) bh_launch_swd_group_adapt(a, grid, block, lds, stream, redundant, (int)wave_lds, FM, PR, cntb) /* swd_group_adapt.hip */
    // The builds without the counters exist for the SIMPLE launches only: there they are worth 2 % (c2 3.44 -> 3.37 ms; 4 instead
    // of 33 spilled SGPRs); a launch with group-velocity targets is 2 % SLOWER without them (c2g 5.55 -> 5.67 ms).
    // the fast arithmetic (FA, see the kernel): every target of the launch takes the short refinement, the usual targets
    const bool farith = a.farith != 0 && build == 2 && simple && wpb == GROUP_WPB;
    a.farith = farith ? 1 : 0;
    if (info != nullptr) info->fast_arith = a.farith;
    if (farith) {
        bh_launch_swd_group_fa(a, grid, block, lds, stream, redundant, (int)wave_lds, adapt, counted, cntb);
    } else if (adapt && build == 2) {
        if (counted) BH_GROUP_LAUNCH_ADAPT(2, true);
        else BH_GROUP_LAUNCH_ADAPT(2, false);
    } else if (adapt && build == 1) { // (both sequences in one launch: Rayleigh targets short, Love targets the reference's)
        // (the instrumented build with both sequences AND the counted scan needs more than 256 registers: it is compiled with
        //  the one-wavefront-per-SIMD budget of swd_group_big.hip -- counters and clocks are what it is for, not speed)
        if (counted && cntb) bh_launch_swd_group_big(a, grid, block, lds, stream, redundant, (int)wave_lds);
        else if (counted) bh_launch_swd_group_adapt(a, grid, block, lds, stream, redundant, (int)wave_lds, 1, true, false);
        else BH_GROUP_LAUNCH_ADAPT(1, false);
    } else if (adapt) {
        if (a.rerun && tun.swd_rerun_wgs > 0 && grid.x > (unsigned)tun.swd_rerun_wgs) grid.x = (unsigned)tun.swd_rerun_wgs; // (the kernel's PASSES: its workgroups stride over the list)
        if (counted) BH_GROUP_LAUNCH_ADAPT(0, true);
        else BH_GROUP_LAUNCH_ADAPT(0, false);
    } else if (build == 2 && simple) {
        if (counted) BH_GROUP_LAUNCH(GROUP_WPB, 2, true, true);
        else BH_GROUP_LAUNCH(GROUP_WPB, 2, true, false);
    } else if (build == 2) {
        BH_GROUP_LAUNCH(GROUP_WPB, 2, false, true);
    } else if (build == 1) {
        BH_GROUP_LAUNCH_(GROUP_WPB, 1, false, true, false, false);
    } else if (simple) {
        if (counted) BH_GROUP_LAUNCH(GROUP_WPB, 0, true, true);
        else BH_GROUP_LAUNCH(GROUP_WPB, 0, true, false);
    } else {
        BH_GROUP_LAUNCH(GROUP_WPB, 0, false, true);
    }
#undef BH_GROUP_LAUNCH_ADAPT
#undef BH_GROUP_LAUNCH
#undef BH_GROUP_LAUNCH_
    return 0;
}
#endif // BH_GROUP_FA_TU
