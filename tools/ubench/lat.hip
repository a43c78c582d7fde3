// Latency micro-benchmarks for the dependent-chain model of the dispersion kernel (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
__global__ void k(double *out, long long *cyc, double a, double b)
{
    __shared__ double lds[64 * 8];
    const int lane = threadIdx.x;
    double x = a + lane * 1e-9, y = b;
    long long t0, t1;
    // 0: dependent fma
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, y);
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[0] = t1 - t0;
    // 1: dependent add
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) x = x + y;
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[1] = t1 - t0;
    // 2: dependent mul
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) x = x * y;
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[2] = t1 - t0;
    // 3: dependent rcp
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x);
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[3] = t1 - t0;
    // 4: dependent division
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) x = y / x;
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[4] = t1 - t0;
    // 5: LDS write -> read (other lane) round trip
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) { lds[lane] = x; __syncthreads(); x = lds[lane ^ 1]; __syncthreads(); }
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[5] = t1 - t0;
    // 6: bpermute of a double
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) x = __shfl(x, lane ^ 1);
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[6] = t1 - t0;
    // 7: dependent fmax
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) x = fmax(x, y) ;
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[7] = t1 - t0;
    // 8: 4 independent fma chains (ILP)
    double p = x, q = x + 1, r = x + 2, s = x + 3;
    asm volatile("s_nop 0" : "+v"(p), "+v"(q), "+v"(r), "+v"(s)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) { p = __builtin_fma(p, y, y); q = __builtin_fma(q, y, y); r = __builtin_fma(r, y, y); s = __builtin_fma(s, y, y); }
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[8] = t1 - t0;
    x = p + q + r + s;
    // 9: sqrt
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < 64; ++i) x = sqrt(x + 2.0);
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[9] = (t1 - t0) * 4;
    // 10: sincos
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < 64; ++i) { double sn, cs; sincos(x + 3.0, &sn, &cs); x = sn + cs; }
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[10] = (t1 - t0) * 4;
    // 11: exp
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < 64; ++i) x = exp(-fabs(x) - 0.5);
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[11] = (t1 - t0) * 4;
    // 12: v_cmp + cndmask style max
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) { if (y > x) x = y * 1.0000001; else x = x + 1e-3; }
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[12] = t1 - t0;
    // 13: ballot + uniform branch
    asm volatile("s_nop 0" : "+v"(x)); t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) { if (__ballot(x > 1e300) == 0ull) x = x + y; else x = x * y; }
    asm volatile("s_nop 0" : "+v"(x)); t1 = clock64(); if (lane == 0) cyc[13] = t1 - t0;
    out[lane] = x;
}
int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 0.99999);
    hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[] = {"fma", "add", "mul", "rcp", "div", "lds w->r", "bpermute f64", "fmax", "4x fma ILP (per 4)", "sqrt", "sincos", "exp", "cmp+select", "ballot+branch+add"};
    for (int i = 0; i < 14; ++i) printf("%-22s %7.1f cycles/op\n", nm[i], h[i] / 256.0);
    return 0;
}
