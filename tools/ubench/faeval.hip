// tools/ubench/faeval.hip -- cycles of ONE secular evaluation, all layers serial in a lane (10-layer model): the reference-exact
// functions of swd_common.h against the fast arithmetic of swd_fa.h, one wavefront per SIMD and two.  Dev tool.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/faeval.hip -o tools/ubench/faeval && tools/ubench/faeval
#include "../../include/bh_engine.h"
#include "../../bayhunter_amd/csrc/bh_device.h"
#include <cstdio>
#define BH_HD __device__ __forceinline__
#define BH_TAB static __device__ const
#include "../../bayhunter_amd/csrc/bh_libm.h"
namespace {
#include "../../bayhunter_amd/csrc/swd_common.h"
}
constexpr int L = 10, NEV = 24;
template <int MODE> // 0 exact Rayleigh, 1 FA Rayleigh, 2 exact Love, 3 FA Love
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, double c0)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const LibmTabs LT = stage_libm_tables(smem, threadIdx.x, blockDim.x);
    float *mdl = reinterpret_cast<float *>(smem + ((LIBM_TAB_BYTES + 15) & ~15)) + (size_t)wave * 4 * L * 64;
    for (int l = 0; l < L; ++l) {
        const float vs = 2.0f + 0.28f * l + 0.001f * lane;
        mdl[(0 * L + l) * 64 + lane] = (l < L - 1) ? 3.0f + 0.5f * l : 0.0f;
        mdl[(1 * L + l) * 64 + lane] = 1.75f * vs;
        mdl[(2 * L + l) * 64 + lane] = vs;
        mdl[(3 * L + l) * 64 + lane] = 0.32f * 1.75f * vs + 0.77f;
    }
    __syncthreads();
    ModelLds md;
    md.d = mdl + lane; md.a = mdl + L * 64 + lane; md.b = mdl + 2 * L * 64 + lane; md.rho = mdl + 3 * L * 64 + lane;
    double c = c0 + 0.003 * lane, acc = 0.0;
    const double omega = 6.283185307179586 / 12.0;
    const long long t0 = clock64();
    for (int i = 0; i < NEV; ++i) {
        DivRange dr; dr.reset();
        double del;
        if (MODE == 0) del = rayleigh_secular<false>(omega / c, omega, md, L, 1, L, dr, LT);
        else if (MODE == 1) del = fa::rayleigh_secular(omega / c, omega, md, L, 1, L);
        else if (MODE == 2) del = love_secular<false>(omega / c, omega, md, L, 1, L, dr, LT);
        else del = fa::love_secular(omega / c, omega, md, L, 1, L);
        acc += del;
        c += 0.005 + 1e-12 * del;
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = (t1 - t0) / NEV;
}
template <int MODE> void run(const char *nm, double *out, long long *cyc)
{
    const size_t lds = ((LIBM_TAB_BYTES + 15) & ~15) + 8 * 4 * L * 64 * sizeof(float);
    for (int wpb : {4, 8}) { // 4 wavefronts of a workgroup: one per SIMD of its CU; 8: two per SIMD
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * wpb), lds, 0, out, cyc, 3.0);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        long long mx = 0; for (int i = 0; i < wpb; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("%-28s %d wavefront(s) per SIMD: %7lld cycles per evaluation (%5.0f per layer)\n", nm, wpb / 4, mx, mx / 9.0);
    }
}
int main()
{
    double *out; long long *cyc; hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 8 * 8);
    run<0>("Rayleigh, reference-exact", out, cyc);
    run<1>("Rayleigh, fast arithmetic", out, cyc);
    run<2>("Love, reference-exact", out, cyc);
    run<3>("Love, fast arithmetic", out, cyc);
    return 0;
}
