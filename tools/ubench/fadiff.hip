// tools/ubench/fadiff.hip -- how far the fast arithmetic's secular value (swd_fa.h) is from the reference-exact one (swd_common.h),
// both max-norm scaled (|f| <= 1): random 3..12-layer models (a quarter with a low-velocity layer), periods 2..60 s, trial
// velocities across the search range including points within 1e-6 ... 1e-12 relative of a layer velocity.  Dev tool; what
// fa::SIGN_FLOOR rests on.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/fadiff.hip -o tools/ubench/fadiff && tools/ubench/fadiff
#include "../../include/bh_engine.h"
#include "../../bayhunter_amd/csrc/bh_device.h"
#include <cstdio>
#include <cmath>
#define BH_HD __device__ __forceinline__
#define BH_TAB static __device__ const
#include "../../bayhunter_amd/csrc/bh_libm.h"
namespace {
#include "../../bayhunter_amd/csrc/swd_common.h"
}
constexpr int L = 12;
__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ float u01(unsigned &s) { s = hash(s + 0x9e3779b9u); return (s >> 8) * (1.0f / 16777216.0f); }
template <int IFUNC>
__global__ __launch_bounds__(64) void k(unsigned long long *hist, double *worst, unsigned seed0)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const LibmTabs LT = stage_libm_tables(smem, threadIdx.x, 64);
    float *mdl = reinterpret_cast<float *>(smem + ((LIBM_TAB_BYTES + 15) & ~15));
    unsigned s = hash(seed0 + 7919u * (blockIdx.x * 64 + lane));
    const int n = 3 + (int)(u01(s) * 9.99f);
    float vsv[L];
    for (int l = 0; l < n; ++l) vsv[l] = 2.0f + 2.8f * u01(s);
    for (int i = 1; i < n; ++i) { float v = vsv[i]; int j = i - 1; while (j >= 0 && vsv[j] > v) { vsv[j + 1] = vsv[j]; --j; } vsv[j + 1] = v; }
    if (u01(s) < 0.25f && n > 3) { const int i = 1 + (int)(u01(s) * (n - 2)); vsv[i] = 0.9f * vsv[i - 1]; }
    const float kk = 1.6f + 0.3f * u01(s);
    for (int l = 0; l < L; ++l) {
        const float v = l < n ? vsv[l] : 1.f;
        mdl[(0 * L + l) * 64 + lane] = (l < n - 1) ? 1.5f + 6.5f * u01(s) : 0.0f;
        mdl[(1 * L + l) * 64 + lane] = v * kk;
        mdl[(2 * L + l) * 64 + lane] = v;
        mdl[(3 * L + l) * 64 + lane] = 0.32f * v * kk + 0.77f;
    }
    __syncthreads();
    ModelLds md;
    md.d = mdl + lane; md.a = mdl + L * 64 + lane; md.b = mdl + 2 * L * 64 + lane; md.rho = mdl + 3 * L * 64 + lane;
    float bmin = 1e9f, bmax = 0.f;
    for (int l = 0; l < n; ++l) { bmin = fminf(bmin, vsv[l]); bmax = fmaxf(bmax, vsv[l]); }
    double wmax = 0.0;
    unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
    for (int it = 0; it < 400; ++it) {
        const double T = 2.0 + 58.0 * u01(s);
        const double omega = 6.283185307179586 / T;
        double c = 0.8 * bmin + (bmax - 0.8 * bmin) * u01(s);
        if ((it & 3) == 0) { // right next to a layer velocity (S, or P for Rayleigh)
            const int l = (int)(u01(s) * n * 0.999f);
            const double v = (IFUNC == 2 && (it & 4)) ? (double)md.Af(l) : (double)md.Bf(l);
            const double rel = pow(10.0, -6.0 - 6.0 * u01(s)) * (u01(s) < 0.5f ? 1.0 : -1.0);
            c = v * (1.0 + rel);
        }
        DivRange dr; dr.reset();
        double fe, ff;
        if (IFUNC == 2) { fe = rayleigh_secular<true>(omega / c, omega, md, n, 1, n, dr, LT); ff = fa::rayleigh_secular(omega / c, omega, md, n, 1, n); }
        else { fe = love_secular<true>(omega / c, omega, md, n, 1, n, dr, LT); ff = fa::love_secular(omega / c, omega, md, n, 1, n); }
        const double d = fabs(fe - ff);
        if (!(d == d)) { h[5]++; continue; }
        wmax = fmax(wmax, d);
        h[0] += d > 1e-13; h[1] += d > 1e-12; h[2] += d > 1e-11; h[3] += d > 1e-10; h[4] += d > 1e-9;
    }
    for (int i = 0; i < 6; ++i) atomicAdd(hist + i, h[i]);
    // max over the grid (non-negative doubles order like their bit patterns)
    atomicMax(reinterpret_cast<unsigned long long *>(worst), (unsigned long long)__double_as_longlong(wmax));
}
int main()
{
    unsigned long long *hist; double *worst;
    hipMalloc(&hist, 6 * 8); hipMalloc(&worst, 8);
    const size_t lds = ((LIBM_TAB_BYTES + 15) & ~15) + 4 * L * 64 * sizeof(float);
    for (int ifunc = 2; ifunc >= 1; --ifunc) {
        hipMemset(hist, 0, 6 * 8); hipMemset(worst, 0, 8);
        const int blocks = 2048;
        if (ifunc == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), lds, 0, hist, worst, 12345u);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), lds, 0, hist, worst, 54321u);
        hipDeviceSynchronize();
        unsigned long long h[6]; double w;
        hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&w, worst, 8, hipMemcpyDeviceToHost);
        const double tot = (double)blocks * 64 * 400;
        printf("%-8s %.3g evaluations (a quarter within 1e-6..1e-12 of a layer velocity): max |f_fast - f_exact| = %.3e; above 1e-13: %llu, 1e-12: %llu, 1e-11: %llu, 1e-10: %llu, 1e-9: %llu; NaN: %llu\n",
               ifunc == 2 ? "Rayleigh" : "Love", tot, w, h[0], h[1], h[2], h[3], h[4], h[5]);
    }
    return 0;
}
