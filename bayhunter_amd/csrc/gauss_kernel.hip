// bayhunter_amd/csrc/gauss_kernel.hip -- the Gauss-law quadratic form d^T R^-1 d for a whole batch
// as a dense FP64 contraction on the matrix cores.
//
// Replaces, for BH_LAW_GAUSS targets, `madist = (ydiff.T).dot(c_inv).dot(ydiff)` of
// src/Targets.py:339-340 with c_inv = R^-1 / sigma^2 (:162-173): for the B residual rows
// D[B][n] and the constant R^-1[n][n] (host LAPACK pinv, once per chain, :150-160) it forms
// V = D * R^-1 tile by tile with v_mfma_f64_16x16x4_f64 and folds  Phi_b = sum_j V_bj D_bj  into
// the epilogue.  This is the one dense contraction of the whole path:
// 2*B*n^2 = 8.6 GFLOP at B = 4096, n = 1024 -- the per-model LDS mat-vec it replaces ran at
// 2.8 TFLOP/s (3.0 ms); the FP64 MFMA peak of MI355X is 78.6 TFLOP/s.
//
// Work split: workgroup (bx, by) = 64 models x one slab of columns; 4 waves, wave w owns models
// [16w, 16w+16) and all four 16-column blocks of the current 64-column tile (one A fragment feeds
// four MFMAs).  K is streamed through LDS in tiles of 32, the global loads of the next tile in flight while
// the MFMAs of the current one run (register double buffer).  Column slabs are summed later in a
// fixed order by like_kernel, so the result does not depend on scheduling (no atomics).
#include "bh_device.h"
#include "bh_tuning.h"
#include <cstdlib>

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int KT = 32;  // K tile
constexpr int LDT = 80; // LDS row stride in doubles: 64 + 16 puts consecutive k rows 128 B apart mod 256 B

__global__ __launch_bounds__(256, 2) void gauss_quad_kernel(int B, int n, int ldy, const double *ymod,
                                                            const double *yobs, const double *rinv,
                                                            int nsplit, int cols_per_split, double *partial)
{
    __shared__ double Dt[KT][LDT]; // residuals, [k][model]
    __shared__ double Rt[KT][LDT]; // R^-1 tile, [k][col]
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int m0 = blockIdx.x * 64;
    const int c_begin = blockIdx.y * cols_per_split;
    const int c_end = min(n, c_begin + cols_per_split);
    const int fi = l & 15, fk = l >> 4; // fragment coordinates
    double acc[4] = {0.0, 0.0, 0.0, 0.0};

    // staging coordinates: D tile: thread -> model tid/4, 8 consecutive k; R^-1 tile: thread -> row tid/8, 8 consecutive columns
    const int d_mdl = tid >> 2, d_kq = (tid & 3) * 8, d_gb = m0 + d_mdl;
    const int r_kr = tid >> 3, r_cq = (tid & 7) * 8;
    for (int jt = c_begin; jt < c_end; jt += 64) {
        double4_t c[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) c[b] = double4_t{0.0, 0.0, 0.0, 0.0};
        // software pipeline: the global loads of K tile t+1 are in flight while the MFMAs of tile t run
        double dreg[8], rreg[8];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = k0 + d_kq + i;
                dreg[i] = (d_gb < B && k < n) ? ymod[(size_t)d_gb * ldy + k] - yobs[k] : 0.0;
            }
            const int k = k0 + r_kr;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int col = jt + r_cq + i;
                rreg[i] = (k < n && col < c_end) ? rinv[(size_t)k * n + col] : 0.0;
            }
        };
        fetch(0);
        for (int k0 = 0; k0 < n; k0 += KT) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                Dt[d_kq + i][d_mdl] = dreg[i];
                Rt[r_kr][r_cq + i] = rreg[i];
            }
            __syncthreads();
            if (k0 + KT < n) fetch(k0 + KT);
#pragma unroll
            for (int kk = 0; kk < KT; kk += 4) {
                const double a = Dt[kk + fk][w * 16 + fi];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double bv = Rt[kk + fk][b * 16 + fi];
                    c[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, c[b], 0, 0, 0);
                }
            }
        }
        // epilogue: c[b][r] = V[model 16w + fk + 4r][col jt + 16b + fi]; fold in D of the same entry
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gb = m0 + w * 16 + fk + 4 * r;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int col = jt + b * 16 + fi;
                if (gb < B && col < c_end) acc[r] += c[b][r] * (ymod[(size_t)gb * ldy + col] - yobs[col]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = acc[r];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        const int gb = m0 + w * 16 + fk + 4 * r;
        if (fi == 0 && gb < B) partial[(size_t)gb * nsplit + blockIdx.y] = v;
    }
}


// ---- the large-problem form (round 3): 128 models x 128 columns per workgroup ---------------------------------
// The 64 x 64 tiles above read 1 MB from L2 for every 8.4 MFLOP (every workgroup streams its 64 residual rows and
// its 64 columns of R^-1 over the whole K range): at 4096 x 1024^2 that is 1 GB per launch, 3.5 TB/s at 0.29 ms --
// the kernel was bound by the L2, not by the matrix cores (37-39 % of their peak).  A 128 x 128 tile halves the bytes
// per flop; eight wavefronts (4 x 2: 32 models x 64 columns each, 8 accumulator blocks, one A fragment feeding four
// MFMAs and one B fragment two) keep two wavefronts on every SIMD; K tiles of 32 are double-buffered in LDS with the
// global loads of tile t+1 in flight during the MFMAs of tile t: ONE barrier per tile (the old form: two).
// LDS layouts chosen for the fragment READS (five per four MFMAs): residuals [model][k] with a row pitch of KT + 1
// doubles -- the 16 models of an A fragment fall into 16 different banks --, R^-1 [k][column], 16 consecutive
// doubles per fragment row.  Staging stores: 8-byte stores for the residuals (conflict-free: 17 m + 4 q covers 16
// banks), 16-byte stores for R^-1.
#ifndef BH_GAUSS_KT
#define BH_GAUSS_KT 16
#endif
constexpr int BM = 128, BN = 128, KT2 = BH_GAUSS_KT;
constexpr int NPT = BM * KT2 / 512; // doubles of each tile a thread stages per K tile (4 at KT2 = 16, 8 at 32)
constexpr int PA = KT2 + 1;  // row pitch of the residual tile (doubles)
constexpr int PB = BN + 4;   // row pitch of the R^-1 tile (doubles; keeps 16-byte alignment)

__global__ __launch_bounds__(512) void gauss_quad_kernel_128(int B, int n, int ldy, const double *__restrict__ ymod,
                                                             const double *__restrict__ yobs, const double *__restrict__ rinv,
                                                             int nsplit, int kper, double *__restrict__ partial)
{
    __shared__ __align__(16) double Dm[2][BM * PA];
    __shared__ __align__(16) double Rt[2][KT2 * PB];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int wm = w >> 1, wn = w & 1;
    const int fi = l & 15, fk = l >> 4;
    const int m0 = blockIdx.x * BM, c0 = blockIdx.y * BN;
    // staging coordinates
    const int d_mdl = tid >> 2, d_kq = (tid & 3) * NPT;                           // residuals: model, NPT consecutive k
    const int r_row = tid / (BN / NPT), r_cq = (tid % (BN / NPT)) * NPT;          // R^-1: k row, NPT consecutive columns
    const int d_gb = m0 + d_mdl;
    const bool d_ok = d_gb < B;
    const double *yrow = ymod + (size_t)(d_ok ? d_gb : 0) * ldy;
    // this workgroup's share of K (blockIdx.z): the quadratic form is a sum over k as well, so a K range is one more slab
    const int kbeg = blockIdx.z * kper, kend = min(n, kbeg + kper);
    double dreg[NPT], rreg[NPT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int k = k0 + d_kq + i;
            dreg[i] = (d_ok && k < kend) ? yrow[k] - yobs[k] : 0.0;
        }
        const int k = k0 + r_row;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int col = c0 + r_cq + i;
            rreg[i] = (k < kend && col < n) ? rinv[(size_t)k * n + col] : 0.0;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) Dm[buf][d_mdl * PA + d_kq + i] = dreg[i];
        double2 *dst = reinterpret_cast<double2 *>(&Rt[buf][r_row * PB + r_cq]);
#pragma unroll
        for (int i = 0; i < NPT / 2; ++i) dst[i] = make_double2(rreg[2 * i], rreg[2 * i + 1]);
    };
    double4_t c[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) c[rb][cb] = double4_t{0.0, 0.0, 0.0, 0.0};
    const int ntile = (kend - kbeg + KT2 - 1) / KT2;
    fetch(kbeg);
    stage(0);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntile) fetch(kbeg + (t + 1) * KT2);
        const double *da = &Dm[buf][(wm * 32 + fi) * PA + fk];
        const double *rb_ = &Rt[buf][fk * PB + wn * 64 + fi];
#pragma unroll
        for (int kk = 0; kk < KT2; kk += 4) {
            const double a0 = da[kk], a1 = da[16 * PA + kk];
            double bv[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) bv[cb] = rb_[kk * PB + cb * 16];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                c[0][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv[cb], c[0][cb], 0, 0, 0);
                c[1][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv[cb], c[1][cb], 0, 0, 0);
            }
        }
        if (t + 1 < ntile) stage(buf ^ 1);
        __syncthreads();
    }
    // epilogue: c[rb][cb][r] = V[model m0 + 32 wm + 16 rb + fk + 4 r][column c0 + 64 wn + 16 cb + fi]; fold in D of the same
    // entry, sum over this wavefront's 64 columns; one slab per (column block, column half)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gb = m0 + wm * 32 + rb * 16 + fk + 4 * r;
            double v = 0.0;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int col = c0 + wn * 64 + cb * 16 + fi;
                if (gb < B && col < n) v += c[rb][cb][r] * (ymod[(size_t)gb * ldy + col] - yobs[col]);
            }
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if (fi == 0 && gb < B) partial[(size_t)gb * nsplit + (blockIdx.y * 2 + wn) * gridDim.z + blockIdx.z] = v;
        }
}

} // namespace

// the 128 x 128 form pays where its grid fills the chip: at least one workgroup for every second CU
static bool use_big_tiles(int B, int n)
{
    const int force = bh_tuning().gauss_tile; // (bh_tuning.h: 64 / 128)
    if (force == 64) return false;
    if (force == 128) return true;
    return (long)((B + BM - 1) / BM) * ((n + BN - 1) / BN) >= 128;
}

// K ranges per tile of the 128 x 128 form: enough workgroups for two per CU (one per CU leaves two wavefronts per SIMD,
// measured 47 % of the matrix peak at 4096 x 1024^2 against 56 % with two)
static int big_ksplit(int B, int n)
{
    const long wgs = (long)((B + BM - 1) / BM) * ((n + BN - 1) / BN);
    int ks = 1;
    while (ks < 4 && wgs * ks < 512 && n / (2 * ks) >= 128) ks *= 2;
    return ks;
}

int bh_gauss_nsplit(int B, int n)
{
    if (use_big_tiles(B, n)) return 2 * ((n + BN - 1) / BN) * big_ksplit(B, n);
    const int tiles = (n + 63) / 64;
    int nsplit = 1;
    // enough workgroups for two per CU, but never less than one 64-column tile per slab
    while (nsplit < tiles && ((B + 63) / 64) * nsplit < 512) nsplit *= 2;
    if (nsplit > tiles) nsplit = tiles;
    return nsplit;
}

void bh_launch_gauss_quad(int B, int n, int ldy, const double *ymod, const double *yobs,
                          const double *rinv, int nsplit, double *partial, hipStream_t stream)
{
    if (use_big_tiles(B, n)) {
        const int ks = big_ksplit(B, n);
        const int kper = (((n + ks - 1) / ks + KT2 - 1) / KT2) * KT2;
        hipLaunchKernelGGL(gauss_quad_kernel_128, dim3((B + BM - 1) / BM, (n + BN - 1) / BN, ks), dim3(512), 0, stream, B, n, ldy,
                           ymod, yobs, rinv, nsplit, kper, partial);
        return;
    }
    const int tiles = (n + 63) / 64;
    const int cols_per_split = ((tiles + nsplit - 1) / nsplit) * 64;
    hipLaunchKernelGGL(gauss_quad_kernel, dim3((B + 63) / 64, nsplit), dim3(256), 0, stream, B, n, ldy, ymod,
                       yobs, rinv, nsplit, cols_per_split, partial);
}
