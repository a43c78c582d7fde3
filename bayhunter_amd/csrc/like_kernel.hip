// bayhunter_amd/csrc/like_kernel.hip -- noise-covariance laws + Gaussian log-likelihood on gfx950.
//
// Replaces, for a batch of models, the per-target part of JointTarget.evaluate
// (src/Targets.py:322-347): RMS misfit (:99-103), the four covariance laws (:105-173) and
//   logL_t = -1/2 (n ln 2pi + ln|C|) - 1/2 d^T C^-1 d                      (:339-342)
// The reference builds a dense n x n inverse covariance on every call (8 MB at n = 1024);
// here the quadratic forms are evaluated in closed form, O(n) per model:
//   nocorr            Phi = sum d_i^2 / sigma^2
//   nocorr_scalederr  Phi = sum d_i^2 / (s_i sigma^2),  s = yerr/min(yerr) (not squared, as there)
//   exponential       Phi = [(1+r^2) sum d_i^2 - r^2 (d_0^2 + d_{n-1}^2) - 2r sum d_i d_{i+1}]
//                            / (sigma^2 (1-r^2))         (tridiagonal inverse of r^|i-j|)
//   gauss (fixed r)   Phi = d^T R^-1 d / sigma^2, R^-1 constant (host LAPACK, once per chain)
// One 256-thread workgroup per model loops over the targets; wave-level shuffles + LDS for the
// reductions.  HBM traffic: the ymod row of the model (n * 8 B per target).
#include "bh_device.h"
#define BH_HD __device__ __forceinline__
#define BH_TAB static __device__ const
#include "bh_libm.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double *red)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void like_kernel(LikeKernelArgs A)
{
    __shared__ double red[4];
    extern __shared__ __align__(16) unsigned char smem[];
    double *dl = reinterpret_cast<double *>(smem); // [max n of the Gauss-law targets]
    const int ib = blockIdx.x;
    const int tid = threadIdx.x;
    const double *y = A.ymod + (size_t)ib * A.ldy;
    double logL = 0.0, joint = 0.0;
    bool failed = false;
    for (int t = 0; t < A.nt; ++t) failed = failed || (A.err_t[(size_t)t * A.B + ib] != 0);
    for (int t = 0; t < A.nt && !failed; ++t) {
        const LikeTargetDev T = A.t[t];
        const int n = T.n;
        const double *ym = y + T.off;
        const double corr = A.noise[(size_t)ib * 2 * A.nt + 2 * t];
        const double sigma = A.noise[(size_t)ib * 2 * A.nt + 2 * t + 1];
        double s0 = 0.0, s1 = 0.0, sw = 0.0;
        double d0 = 0.0, dn = 0.0;
        if (T.pre != nullptr) { // the forward kernel formed the sums (fused likelihood, RfKernelArgs::sums): nothing to read of ymod
            const double *pre = T.pre + (size_t)ib * 4;
            s0 = pre[0];
            s1 = pre[1];
            d0 = pre[2];
            dn = pre[3];
        } else {
        for (int i = tid; i < n; i += 256) {
            const double d = ym[i] - T.yobs[i];
            s0 += d * d;
            if (T.law == 2 && i + 1 < n) s1 += d * (ym[i + 1] - T.yobs[i + 1]);
            if (T.law == 1) sw += d * d / T.yerr_scaled[i];
        }
        if (T.law == 3 && T.quad != nullptr) { // slab sums from the MFMA contraction, fixed order
            if (tid == 0)
                for (int sidx = 0; sidx < T.nsplit; ++sidx) sw += T.quad[(size_t)ib * T.nsplit + sidx];
        } else if (T.law == 3) { // (d^T R^-1) d with d staged in LDS; column access = coalesced over i
            __syncthreads();
            for (int i = tid; i < n; i += 256) dl[i] = ym[i] - T.yobs[i];
            __syncthreads();
            for (int i = tid; i < n; i += 256) {
                double acc = 0.0;
                for (int jj = 0; jj < n; ++jj) acc += dl[jj] * T.rinv[(size_t)jj * n + i];
                sw += acc * dl[i];
            }
        }
        if (T.law == 2) {
            d0 = ym[0] - T.yobs[0];
            dn = ym[n - 1] - T.yobs[n - 1];
        }
        }
        if (T.pre != nullptr) {
            // (already reduced)
        } else if (n <= 64 && !(T.law == 3 && T.quad == nullptr)) {
            // a short target (a dispersion curve beside a long receiver function): its samples all sit in the first
            // wavefront, the other three would only add zeros -- no barrier (same bits as block_sum); only thread 0's
            // values are used below
            for (int off = 32; off > 0; off >>= 1) s0 += __shfl_xor(s0, off);
            if (T.law == 2)
                for (int off = 32; off > 0; off >>= 1) s1 += __shfl_xor(s1, off);
            if (T.law == 1 || T.law == 3)
                for (int off = 32; off > 0; off >>= 1) sw += __shfl_xor(sw, off);
        } else {
            s0 = block_sum(s0, red);
            if (T.law == 2) s1 = block_sum(s1, red);
            if (T.law == 1 || T.law == 3) sw = block_sum(sw, red);
        }
        const double s2 = sigma * sigma;
        double phi, logdet = (2.0 * n) * log(sigma);
        if (T.law == 0) {
            phi = s0 / s2;
        } else if (T.law == 1) {
            phi = sw / s2;
            logdet += T.logdet_extra;
        } else if (T.law == 2) {
            // get_corr_inv (Targets.py:131-137): d[0] = d[-1] = 1 -- for n == 1 both hit the
            // same element, so the edge correction must not be applied twice
            const double edge = (n > 1) ? (d0 * d0 + dn * dn) : (d0 * d0);
            const double r2 = corr * corr;
            phi = ((1.0 + r2) * s0 - r2 * edge - 2.0 * corr * s1) / (s2 * (1.0 - r2));
            logdet += (n - 1) * log(1.0 - r2);
        } else {
            phi = sw / s2;
            logdet += T.logdet_extra;
        }
        const double part = -0.5 * ((double)n * log(2.0 * M_PI) + logdet);
        logL += part - phi / 2.0;
        const double rms = sqrt(s0 / (double)n);
        joint += rms;
        if (tid == 0) A.misfits[(size_t)ib * (A.nt + 1) + t] = rms;
    }
    if (tid == 0) {
        if (failed) { // Targets.py:325-328
            A.logL[ib] = -1e15;
            for (int t = 0; t <= A.nt; ++t) A.misfits[(size_t)ib * (A.nt + 1) + t] = 1e15;
            A.err[ib] = 1;
        } else {
            A.logL[ib] = logL;
            A.misfits[(size_t)ib * (A.nt + 1) + A.nt] = joint;
            A.err[ib] = 0;
        }
    }
}

// All targets short (n <= 64, e.g. dispersion curves): one WAVEFRONT per model, four models per workgroup, no LDS and
// no barrier -- the workgroup-per-model form above spends most of its 37 us (B = 4096, two targets of 30 periods) in
// two barriers per reduction.  Lane i holds sample i, the sums are the same xor-shuffle tree over the same 64 slots
// (the other three wavefronts of the form above only add zeros): identical bits.
__global__ __launch_bounds__(256) void like_small_kernel(LikeKernelArgs A)
{
    const int lane = threadIdx.x & 63;
    const int ib = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ib >= A.B) return;
    const double *y = A.ymod + (size_t)ib * A.ldy;
    double logL = 0.0, joint = 0.0;
    bool failed = false;
    for (int t = 0; t < A.nt; ++t) failed = failed || (A.err_t[(size_t)t * A.B + ib] != 0);
    for (int t = 0; t < A.nt && !failed; ++t) {
        const LikeTargetDev T = A.t[t];
        const int n = T.n;
        const double *ym = y + T.off;
        const double corr = A.noise[(size_t)ib * 2 * A.nt + 2 * t];
        const double sigma = A.noise[(size_t)ib * 2 * A.nt + 2 * t + 1];
        double s0 = 0.0, s1 = 0.0, sw = 0.0, d0 = 0.0, dn = 0.0;
        if (T.pre != nullptr) { // the forward kernel formed the sums (fused likelihood, RfKernelArgs::sums)
            const double *pre = T.pre + (size_t)ib * 4;
            s0 = pre[0];
            s1 = pre[1];
            d0 = pre[2];
            dn = pre[3];
        } else {
        if (lane < n) {
            const double d = ym[lane] - T.yobs[lane];
            s0 += d * d;
            if (T.law == 2 && lane + 1 < n) s1 += d * (ym[lane + 1] - T.yobs[lane + 1]);
            if (T.law == 1) sw += d * d / T.yerr_scaled[lane];
        }
        if (T.law == 3 && lane == 0)
            for (int sidx = 0; sidx < T.nsplit; ++sidx) sw += T.quad[(size_t)ib * T.nsplit + sidx];
        for (int off = 32; off > 0; off >>= 1) s0 += __shfl_xor(s0, off);
        if (T.law == 2)
            for (int off = 32; off > 0; off >>= 1) s1 += __shfl_xor(s1, off);
        if (T.law == 1 || T.law == 3)
            for (int off = 32; off > 0; off >>= 1) sw += __shfl_xor(sw, off);
        if (T.law == 2) {
            d0 = ym[0] - T.yobs[0];
            dn = ym[n - 1] - T.yobs[n - 1];
        }
        }
        const double s2 = sigma * sigma;
        double phi, logdet = (2.0 * n) * log(sigma);
        if (T.law == 0) {
            phi = s0 / s2;
        } else if (T.law == 1) {
            phi = sw / s2;
            logdet += T.logdet_extra;
        } else if (T.law == 2) {
            const double edge = (n > 1) ? (d0 * d0 + dn * dn) : (d0 * d0);
            const double r2 = corr * corr;
            phi = ((1.0 + r2) * s0 - r2 * edge - 2.0 * corr * s1) / (s2 * (1.0 - r2));
            logdet += (n - 1) * log(1.0 - r2);
        } else {
            phi = sw / s2;
            logdet += T.logdet_extra;
        }
        const double part = -0.5 * ((double)n * log(2.0 * M_PI) + logdet);
        logL += part - phi / 2.0;
        const double rms = sqrt(s0 / (double)n);
        joint += rms;
        if (lane == 0) A.misfits[(size_t)ib * (A.nt + 1) + t] = rms;
    }
    if (lane == 0) {
        if (failed) { // Targets.py:325-328
            A.logL[ib] = -1e15;
            for (int t = 0; t <= A.nt; ++t) A.misfits[(size_t)ib * (A.nt + 1) + t] = 1e15;
            A.err[ib] = 1;
        } else {
            A.logL[ib] = logL;
            A.misfits[(size_t)ib * (A.nt + 1) + A.nt] = joint;
            A.err[ib] = 0;
        }
    }
}

__global__ void probe_kernel(int op, int n, const double *in, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = (op < 6) ? in[i] : 0.0; // ops >= 6 read their own operands
    double r;
    if (op == 6 || op == 7) { // division probes: in = pairs (a, b)
        const double a = in[2 * i], b = in[2 * i + 1];
        out[i] = (op == 6) ? bh_quot(a, b, bh_rcp_refined(b)) : a / b;
        return;
    }
    if (op >= 8 && op <= 10) { // the glibc-exact exp / sincos of bh_libm.h (tables read from global memory)
        const double v = in[i];
        double sn = 0.0, cs = 0.0;
        if (op == 10) {
            out[i] = bhp_exp_in_domain(v) ? bhp_exp_core(v, bhp_exp_tab) : exp(v);
        } else {
            if (!bhp_sincos_bl(v, &sn, &cs, reinterpret_cast<const double *>(bhp_sincos_tab_bits))) sincos(v, &sn, &cs);
            out[i] = (op == 8) ? sn : cs;
        }
        return;
    }
    switch (op) {
    case 0: r = sqrt(x); break;
    case 1: r = sin(x); break;
    case 2: r = cos(x); break;
    case 3: r = exp(x); break;
    case 4: r = log(x); break;
    default: r = 1.0 / x; break;
    }
    out[i] = r;
}

} // namespace

void bh_launch_like(const LikeKernelArgs &a, hipStream_t stream)
{
    size_t lds = 0;
    for (int t = 0; t < a.nt; ++t)
        if (a.t[t].law == 3 && a.t[t].quad == nullptr && (size_t)a.t[t].n * sizeof(double) > lds)
            lds = (size_t)a.t[t].n * sizeof(double);
    // every target fits one wavefront -- or comes with its sums already formed by the forward kernel (a receiver function's fused
    // likelihood: nothing of its trace is read here) -- and a Gauss law has its slab sums from the MFMA contraction
    bool small = true;
    for (int t = 0; t < a.nt; ++t)
        small = small && (a.t[t].n <= 64 || (a.t[t].pre != nullptr && a.t[t].law != 3)) && !(a.t[t].law == 3 && a.t[t].quad == nullptr);
    if (small) hipLaunchKernelGGL(like_small_kernel, dim3((a.B + 3) / 4), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(like_kernel, dim3(a.B), dim3(256), lds, stream, a);
}

void bh_launch_probe(int op, int n, const double *in, double *out, hipStream_t stream)
{
    hipLaunchKernelGGL(probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, n, in, out);
}
