#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#define BH_HD __host__ __device__ static inline
#define BH_TAB static __device__ const
#include "../../bayhunter_amd/csrc/bh_libm.h"
__global__ void k(double x, double *o)
{
    const double *tab = reinterpret_cast<const double *>(bhp_sincos_tab_bits);
    double dx = 0.0;
    if (x < 0) dx = -dx;
    const double ax = fabs(x);
    const double u = BHP_BIG + ax;
    const double x1 = ax - (u - BHP_BIG) + dx;
    const double xx = x1 * x1;
    const double s = x1 + x1 * xx * (BHP_SN3 + xx * BHP_SN5);
    const double c = xx * (BHP_CS2 + xx * (BHP_CS4 + xx * BHP_CS6));
    const int kk = (int)(uint32_t)bhp_asuint(u) * 4;
    const double sn = tab[kk], ssn = tab[kk + 1], cs = tab[kk + 2], ccs = tab[kk + 3];
    const double cor = (ccs - s * ssn - cs * c) - sn * s;
    o[0] = u; o[1] = x1; o[2] = xx; o[3] = s; o[4] = c; o[5] = sn; o[6] = ssn; o[7] = cs; o[8] = ccs; o[9] = cor; o[10] = cs + cor; o[11] = kk;
    double sn2, cs2; bhp_sincos(x, &sn2, &cs2, tab); o[12] = cs2;
}
int main()
{
    double *d; hipMalloc(&d, 16 * 8); double h[16];
    const double x = -0.05806683587376953;
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, x, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[] = {"u", "x1", "xx", "s", "c", "sn", "ssn", "cs", "ccs", "cor", "res", "k", "port_cos"};
    for (int i = 0; i < 13; ++i) printf("%-8s %a  %.17g\n", nm[i], h[i], h[i]);
    double s0, c0; sincos(x, &s0, &c0); printf("libm cos %a\n", c0);
    return 0;
}
