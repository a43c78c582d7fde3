// Micro-benchmark of phase-B (vector recursion) variants of the dispersion kernel (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../bayhunter_amd/csrc/bh_device.h"
#define NL 9
#define G 9
#define MPW 7
#define CAS 26
__device__ __forceinline__ void norm_plain(double ee[5], double e[5]) {
    double t1 = 0.0;
    for (int i = 0; i < 5; ++i) if (fabs(ee[i]) > t1) t1 = fabs(ee[i]);
    if (t1 < 1e-40) t1 = 1.0;
    for (int i = 0; i < 5; ++i) e[i] = ee[i] / t1;
}
template <int CHECK> __device__ __forceinline__ void norm_fast(const double ee[5], double e[5]) {
    const double a0 = fabs(ee[0]), a1 = fabs(ee[1]), a2 = fabs(ee[2]), a3 = fabs(ee[3]), a4 = fabs(ee[4]);
    double t1 = fmax(fmax(fmax(a0, a1), fmax(a2, a3)), a4);
    if (t1 < 1e-40) t1 = 1.0;
    bool safe = true;
    if (CHECK) { const double mn = fmin(fmin(fmin(a0, a1), fmin(a2, a3)), a4); safe = bh_div_safe(t1) && bh_div_safe(mn); }
    if (!CHECK || __ballot(!safe) == 0ull) {
        const double r = bh_rcp_refined(t1);
        for (int i = 0; i < 5; ++i) e[i] = bh_quot(ee[i], t1, r);
    } else for (int i = 0; i < 5; ++i) e[i] = ee[i] / t1;
}
// predicated fallback: fast path always, plain divisions only in lanes that fail the range test
__device__ __forceinline__ void norm_pred(const double ee[5], double e[5]) {
    const double a0 = fabs(ee[0]), a1 = fabs(ee[1]), a2 = fabs(ee[2]), a3 = fabs(ee[3]), a4 = fabs(ee[4]);
    double t1 = fmax(fmax(fmax(a0, a1), fmax(a2, a3)), a4);
    if (t1 < 1e-40) t1 = 1.0;
    const double mn = fmin(fmin(fmin(a0, a1), fmin(a2, a3)), a4);
    const bool safe = bh_div_safe(t1) && bh_div_safe(mn);
    const double r = bh_rcp_refined(t1);
    for (int i = 0; i < 5; ++i) e[i] = bh_quot(ee[i], t1, r);
    if (!safe) for (int i = 0; i < 5; ++i) e[i] = ee[i] / t1;
}
template <int V> __global__ __launch_bounds__(64) void rk(const double *cain, double *out, long long *cyc, int nev)
{
    __shared__ double ca[MPW * NL * CAS];
    __shared__ double ex[MPW * 12];
    const int lane = threadIdx.x, g = lane < 63 ? lane / G : 0, li = lane < 63 ? lane % G : 0, col = li % 5;
    for (int i = lane; i < MPW * NL * CAS; i += 64) ca[i] = cain[i];
    __syncthreads();
    const double *cam = ca + g * NL * CAS; double *exm = ex + g * 12;
    double e[5] = {1.0, -0.3, 0.2, 0.7, -0.1};
    long long t0 = clock64();
    for (int ev = 0; ev < nev; ++ev) {
        for (int m = NL - 1; m >= 0; --m) {
            if (V <= 2) {
                const double *cc = cam + m * CAS; double ee[5];
                for (int i = 0; i < 5; ++i) { double acc = 0.0; for (int j = 0; j < 5; ++j) acc = acc + e[j] * cc[5 * i + j]; ee[i] = acc; }
                if (V == 0) norm_plain(ee, e); else if (V == 1) norm_fast<1>(ee, e); else norm_fast<0>(ee, e);
            } else {
                const double *cc = cam + m * CAS + 5 * col;
                double ee = 0.0; for (int j = 0; j < 5; ++j) ee = ee + e[j] * cc[j];
                if (V == 3) {
                    if (li < 5) exm[col] = ee; __syncthreads();
                    double t1 = 0.0; for (int i = 0; i < 5; ++i) { double a = fabs(exm[i]); if (a > t1) t1 = a; } if (t1 < 1e-40) t1 = 1.0;
                    const double en = ee / t1; if (li < 5) exm[6 + col] = en; __syncthreads();
                    for (int i = 0; i < 5; ++i) e[i] = exm[6 + i];
                } else if (V == 4 || V == 6) {
                    if (li < 5) exm[col] = ee; __syncthreads();
                    double v[5]; for (int i = 0; i < 5; ++i) v[i] = exm[i];
                    if (V == 4) norm_fast<1>(v, e); else norm_fast<0>(v, e);
                } else { // 5/7: bpermute exchange
                    double v[5]; for (int i = 0; i < 5; ++i) v[i] = __shfl(ee, g * G + i);
                    if (V == 5) norm_fast<0>(v, e); else norm_pred(v, e);
                }
            }
        }
    }
    long long t1 = clock64();
    if (lane == 0) cyc[0] = t1 - t0;
    out[lane] = e[0] + e[1] + e[2] + e[3] + e[4];
}
template <int V> __global__ __launch_bounds__(64) void lk(const double *cain, double *out, long long *cyc, int nev)
{
    __shared__ double ca[MPW * NL * CAS];
    const int lane = threadIdx.x, g = lane < 63 ? lane / G : 0;
    for (int i = lane; i < MPW * NL * CAS; i += 64) ca[i] = cain[i];
    __syncthreads();
    const double *cam = ca + g * NL * CAS;
    double e1 = 1.0, e2 = 0.3;
    long long t0 = clock64();
    for (int ev = 0; ev < nev; ++ev)
        for (int m = NL - 1; m >= 0; --m) {
            const double2 *src = reinterpret_cast<const double2 *>(cam + m * CAS);
            const double2 p0 = src[0], p1 = src[1], p2 = src[2];
            const double cosq = p0.x, y = p0.y, z = p1.x, xmu = p1.y, rx = p2.x;
            const double e10 = e1 * cosq + e2 * xmu * z;
            double e20;
            if (V == 0) e20 = e1 * y / xmu + e2 * cosq;
            else if (V == 1) { const double num = e1 * y; const bool s = bh_div_safe(xmu) && bh_div_safe(num);
                double q; if (__ballot(!s) == 0ull) q = bh_quot(num, xmu, rx); else q = num / xmu; e20 = q + e2 * cosq; }
            else if (V == 3) { const double num = e1 * y; double q = bh_quot(num, xmu, rx); if (!(bh_div_safe(xmu) && bh_div_safe(num))) q = num / xmu; e20 = q + e2 * cosq; }
            else e20 = bh_quot(e1 * y, xmu, rx) + e2 * cosq;
            double xnor = fabs(e10); const double yn = fabs(e20); if (yn > xnor) xnor = yn; if (xnor < 1e-40) xnor = 1.0;
            if (V == 0) { e1 = e10 / xnor; e2 = e20 / xnor; }
            else if (V == 1) { const bool s = bh_div_safe(xnor) && bh_div_safe(fmin(fabs(e10), yn));
                if (__ballot(!s) == 0ull) { const double r = bh_rcp_refined(xnor); e1 = bh_quot(e10, xnor, r); e2 = bh_quot(e20, xnor, r); } else { e1 = e10 / xnor; e2 = e20 / xnor; } }
            else if (V == 3) { const bool s = bh_div_safe(xnor) && bh_div_safe(fmin(fabs(e10), yn)); const double r = bh_rcp_refined(xnor);
                double q1 = bh_quot(e10, xnor, r), q2 = bh_quot(e20, xnor, r); if (!s) { q1 = e10 / xnor; q2 = e20 / xnor; } e1 = q1; e2 = q2; }
            else { const double r = bh_rcp_refined(xnor); e1 = bh_quot(e10, xnor, r); e2 = bh_quot(e20, xnor, r); }
        }
    long long t1 = clock64();
    if (lane == 0) cyc[0] = t1 - t0;
    out[lane] = e1 + e2;
}
int main()
{
    const int n = MPW * NL * CAS; double h[n]; srand(1);
    for (int i = 0; i < n; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5) * 2.0 + 0.1;
    for (int m = 0; m < MPW * NL; ++m) { h[m * CAS + 3] = 30.0 + m; h[m * CAS + 4] = 1.0 / h[m * CAS + 3]; } // love: xmu, rx-ish
    double *din, *out; long long *cyc; hipMalloc(&din, n * 8); hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    hipMemcpy(din, h, n * 8, hipMemcpyHostToDevice);
    const int nev = 200; double res[64]; long long c;
#define RUN(K, name) for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, din, out, cyc, nev); hipDeviceSynchronize(); \
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(res, out, 64 * 8, hipMemcpyDeviceToHost); printf("%-46s %7.1f cycles/step  (chk %.17g)\n", name, c / (double)(nev * NL), res[0]);
    RUN(rk<0>, "R0 redundant, plain div, seq max");
    RUN(rk<1>, "R1 redundant, shared rcp + ballot check");
    RUN(rk<2>, "R2 redundant, shared rcp, no check");
    RUN(rk<3>, "R3 par5, 2 LDS exchanges, own plain div");
    RUN(rk<4>, "R4 par5, 1 LDS exchange, shared rcp + check");
    RUN(rk<6>, "R6 par5, 1 LDS exchange, shared rcp, no check");
    RUN(rk<5>, "R5 par5, bpermute exchange, shared rcp no check");
    RUN(rk<7>, "R7 par5, bpermute, shared rcp, predicated fallback");
    RUN(lk<0>, "L0 plain divisions");
    RUN(lk<1>, "L1 shared rcp + ballot checks");
    RUN(lk<2>, "L2 shared rcp, no checks");
    RUN(lk<3>, "L3 shared rcp, predicated fallback");
    return 0;
}
