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

} // namespace

int bh_gauss_nsplit(int B, int n)
{
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
    const int tiles = (n + 63) / 64;
    const int cols_per_split = ((tiles + nsplit - 1) / nsplit) * 64;
    hipLaunchKernelGGL(gauss_quad_kernel, dim3((B + 63) / 64, nsplit), dim3(256), 0, stream, B, n, ldy, ymod,
                       yobs, rinv, nsplit, cols_per_split, partial);
}
