#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define BH_HD __device__ __forceinline__
#define BH_TAB static __device__ const
#include "../../bayhunter_amd/csrc/bh_libm.h"
__global__ void k(double *out, long long *cyc, double a)
{
    __shared__ uint64_t et[256];
    __shared__ uint64_t st[440];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) et[i] = bhp_exp_tab[i];
    for (int i = lane; i < 440; i += 64) st[i] = bhp_sincos_tab_bits[i];
    __syncthreads();
    const double *tab = (const double *)st;
    double x = a + lane * 0.37;
    long long t0, t1;
#define TIC asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#define TOC(i, n) asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[i] = (t1 - t0) / n;
    TIC for (int i = 0; i < 64; ++i) { double s, c; sincos(x + 3.0, &s, &c); x = s + c; } TOC(0, 64)
    TIC for (int i = 0; i < 64; ++i) { double s, c; bhp_sincos_bl(x + 3.0, &s, &c, tab); x = s + c; } TOC(1, 64)
    TIC for (int i = 0; i < 64; ++i) { double s, c; bhp_sincos(x + 3.0, &s, &c, tab); x = s + c; } TOC(2, 64)
    TIC for (int i = 0; i < 64; ++i) { x = exp(-fabs(x) - 0.5); } TOC(3, 64)
    TIC for (int i = 0; i < 64; ++i) { x = bhp_exp_core(-fabs(x) - 0.5, et); } TOC(4, 64)
    TIC for (int i = 0; i < 64; ++i) { x = sqrt(x + 2.0); } TOC(5, 64)
    TIC for (int i = 0; i < 64; ++i) { x = 1.7 / (x + 2.0); } TOC(6, 64)
    out[lane] = x;
}
int main()
{
    double *out; long long *cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 0.3);
    hipDeviceSynchronize(); long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[] = {"ocml sincos", "glibc-port sincos (branch-light)", "glibc-port sincos (branchy, lanes diverge)", "ocml exp", "glibc-port exp", "sqrt", "div"};
    for (int i = 0; i < 7; ++i) printf("%-44s %5lld cycles\n", nm[i], h[i]);
}
