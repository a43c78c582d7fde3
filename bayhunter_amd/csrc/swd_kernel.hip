// bayhunter_amd/csrc/swd_kernel.hip -- Rayleigh/Love phase & group dispersion on gfx950.
//
// Replaces the reference's surfdisp96 (src/extensions/surfdisp96.f:55-360 and the routines it
// calls) for a batch of models.  No MFMA: the work is a scalar FP64 recurrence (5-vector x 5x5
// compound matrix per layer for Rayleigh, 2-vector for Love) inside a data-dependent root search.
// Two kernels return identical bits:
//   swd_kernel<1|2>     one wavefront lane = one model (batches that fill the chip by themselves);
//   swd_group_kernel    G lanes = one model, J such groups evaluating the trial velocities the search
//                       will most probably ask for next (everything smaller) -- see the block comment
//                       above it; bh_swd_plan picks the mapping and the look-ahead per launch.
//
// Design points
//   * EVALUATION-SYNCHRONOUS STATE MACHINE.  The reference's control flow is
//       for period: bracket-step until sign change; refine (bisection / inverse Neville)
//     and the number of secular-function evaluations differs per model and per period.  A
//     literal SIMT translation would make every lane wait for the slowest lane in every inner
//     loop.  Here each model keeps an explicit search state (period index, which root, bracket,
//     Neville table, continuation tag) and the wavefront's loop body is exactly ONE secular
//     evaluation for all its models followed by a state transition.  Models drift apart in
//     period index freely; the wave ends when its slowest model has used up its own total, not
//     the sum of per-period maxima.
//   * The model (thickness, vp, vs, rho), rounded to binary32 as the f2py boundary of the
//     reference does, is staged once through LDS from coalesced global loads
//     of the layer-major (vp, vs, rho, h) arrays; the Neville tables x[11], y[11] and the period
//     table live in LDS as well.
//   * Rounding points, the search sequence (start value, 0.005 km/s stepping, direction logic,
//     Neville/bisection decisions, the 1e-6 stop test, which point is returned), the binary32
//     arithmetic of the start value and of the group-velocity formula follow the reference exactly
//     (SURVEY.md App. A), and sin/cos/exp (and log/powf of the flattening transform) are
//     restatements of the host libm the reference links (bh_libm.h): results are bit-identical.
#include "../../include/bh_engine.h"
#include "bh_device.h"
#include <cstdlib>

// glibc-exact exp / sincos (see bh_libm.h): with these the device's secular function is the
// reference's bit for bit -- sqrt and division are correctly rounded on gfx950, and exp / sincos
// were the only operations where the device library (ocml, <= 1 ulp) and the host libm differed.
#define BH_HD __device__ __forceinline__
#define BH_TAB static __device__ const
#include "bh_libm.h"

namespace {

struct LibmTabs {
    const uint64_t *exp_tab; // [256]  in LDS
    const double *sc_tab;    // [440]  in LDS
};
constexpr int LIBM_TAB_BYTES = 256 * 8 + 440 * 8;

__device__ __forceinline__ LibmTabs stage_libm_tables(unsigned char *lds, int lane, int nthreads = BH_WAVE)
{
    uint64_t *et = reinterpret_cast<uint64_t *>(lds);
    uint64_t *st = et + 256;
    for (int i = lane; i < 256; i += nthreads) et[i] = bhp_exp_tab[i];
    for (int i = lane; i < 440; i += nthreads) st[i] = bhp_sincos_tab_bits[i];
    return LibmTabs{et, reinterpret_cast<const double *>(st)};
}
__device__ __forceinline__ void bh_sincos(double x, double *sn, double *cs, const LibmTabs &T)
{
    if (!bhp_sincos_bl(x, sn, cs, T.sc_tab)) sincos(x, sn, cs); // |x| >= 1.05e8, inf, nan: device library
}
__device__ __forceinline__ double bh_exp(double x, const LibmTabs &T)
{
    double r = bhp_exp_core(x, T.exp_tab); // branch-free main path; meaningless outside its domain
    if (!bhp_exp_in_domain(x))             // rare: |x| < 2^-54 -> 1 + x like glibc; |x| >= 512, nan -> device library
        r = ((((unsigned)__double2hiint(x) >> 20) & 0x7ffu) < 0x3c9u) ? 1.0 + x : exp(x);
    return r;
}


constexpr int NEV_MAX = 11; // Neville table entries: order grows to m <= 10 (surfdisp96.f:655)

__device__ __forceinline__ bool signs_differ(double x, double y)
{
    return ((__double_as_longlong(x) ^ __double_as_longlong(y)) < 0);
}

// LDS views -----------------------------------------------------------------------------------
// Model arrays in LDS as [array][layer][column]; S = number of columns (models) per wave.
template <int S>
struct ModelLdsT {
    const float *d, *a, *b, *rho; // column pre-offset
    __device__ __forceinline__ float Df(int m) const { return d[m * S]; }
    __device__ __forceinline__ float Af(int m) const { return a[m * S]; }
    __device__ __forceinline__ float Bf(int m) const { return b[m * S]; }
    __device__ __forceinline__ double D(int m) const { return (double)d[m * S]; }
    __device__ __forceinline__ double A(int m) const { return (double)a[m * S]; }
    __device__ __forceinline__ double Bv(int m) const { return (double)b[m * S]; }
    __device__ __forceinline__ double R(int m) const { return (double)rho[m * S]; }
};
using ModelLds = ModelLdsT<BH_WAVE>;
struct ModelLdsRt { // same, with the column count known only at run time
    const float *d, *a, *b, *rho;
    int S;
    __device__ __forceinline__ float Df(int m) const { return d[m * S]; }
    __device__ __forceinline__ float Af(int m) const { return a[m * S]; }
    __device__ __forceinline__ float Bf(int m) const { return b[m * S]; }
    __device__ __forceinline__ double D(int m) const { return (double)d[m * S]; }
    __device__ __forceinline__ double A(int m) const { return (double)a[m * S]; }
    __device__ __forceinline__ double Bv(int m) const { return (double)b[m * S]; }
    __device__ __forceinline__ double R(int m) const { return (double)rho[m * S]; }
};

// ---- range tracking for the shared-reciprocal divisions ------------------------------------------
// The fast division route (bh_device.h) returns the bits of a plain IEEE division as long as the
// operands lie in [2^-400, 2^400].  Checking that with a branch inside the layer recursion costs
// more than it saves (an exec-mask branch is ~100 cycles on this chip), so the recursion only
// TRACKS the smallest and largest magnitude it divided (two cheap min/max per operand, no
// branch); after the whole recursion one test decides whether the value can be trusted, and the
// (never observed in practice) out-of-range case re-runs the recursion with plain divisions.
struct DivRange {
    double lo, hi;
    __device__ __forceinline__ void reset() { lo = 1.0; hi = 1.0; }
    __device__ __forceinline__ void see(double a) { lo = fmin(lo, a); hi = fmax(hi, a); } // a >= 0
    __device__ __forceinline__ bool ok() const
    {
        return lo >= 3.8725919148493183e-121 /* 2^-400 */ && hi <= 2.5822498780869086e+120 /* 2^400 */;
    }
};

// One layer of the Love recursion (surfdisp96.f:758-767) given the layer terms.
// EXACT = true: the reference's operations verbatim.  EXACT = false: the three divisions take
// the shared-reciprocal route (rx = bh_rcp_refined(xmu)) and report their operand range.
template <bool EXACT>
__device__ __forceinline__ void love_step(double &e1, double &e2, double cosq, double y, double z,
                                          double xmu, double rx, DivRange &dr)
{
    const double e10 = e1 * cosq + e2 * xmu * z;
    double e20;
    if (EXACT) {
        e20 = e1 * y / xmu + e2 * cosq;
    } else {
        const double num = e1 * y;
        dr.see(fabs(num));
        dr.see(xmu);
        e20 = bh_quot(num, xmu, rx) + e2 * cosq;
    }
    const double a10 = fabs(e10), a20 = fabs(e20);
    double xnor = fmax(a10, a20);
    if (xnor < 1.0e-40) xnor = 1.0;
    if (EXACT) {
        e1 = e10 / xnor;
        e2 = e20 / xnor;
    } else {
        dr.see(fmin(a10, a20));
        dr.see(xnor);
        const double r = bh_rcp_refined(xnor);
        e1 = bh_quot(e10, xnor, r);
        e2 = bh_quot(e20, xnor, r);
    }
}

// ---- Love: SH Thomson-Haskell (surfdisp96.f:710-769) ----------------------------------------
template <bool EXACT>
__device__ double love_secular(double wvno, double omega, const ModelLds &md, int mmax, int llw,
                               int mtop, DivRange &dr, const LibmTabs &LT)
{
    double beta1 = md.Bv(mmax - 1);
    double rho1 = md.R(mmax - 1);
    double xkb = omega / beta1;
    double wvnop = wvno + xkb;
    double wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double e1 = rho1 * rb;
    double e2 = 1.0 / (beta1 * beta1);
    for (int m = mtop - 2; m >= 0; --m) {
        if (m <= mmax - 2 && m >= llw - 1) {
            beta1 = md.Bv(m);
            rho1 = md.R(m);
            const double dm = md.D(m);
            const double xmu = rho1 * beta1 * beta1;
            xkb = omega / beta1;
            wvnop = wvno + xkb;
            wvnom = fabs(wvno - xkb);
            rb = sqrt(wvnop * wvnom);
            const double q = dm * rb;
            double cosq, y, z;
            if (wvno < xkb) {
                double sinq;
                bh_sincos(q, &sinq, &cosq, LT);
                y = sinq / rb;
                z = -rb * sinq;
            } else if (wvno == xkb) {
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
            love_step<EXACT>(e1, e2, cosq, y, z, xmu, EXACT ? 0.0 : bh_rcp_refined(xmu), dr);
        }
    }
    return e1;
}

// ---- Rayleigh: eigenfunction products (surfdisp96.f:874-991, `var`) --------------------------
struct LayerTerms {
    double a0, cpcq, cpy, cpz, cqw, cqx, xy, xz, wy, wz, w, cosp;
};

__device__ __forceinline__ void layer_products(double p, double q, double ra, double rb,
                                               double wvno, double xka, double xkb, double dpth,
                                               LayerTerms &o, const LibmTabs &LT)
{
    // Two exec-mask regions per wave type (propagating: sincos; evanescent: exp) instead of the
    // Fortran's three-way ifs -- branches are the expensive thing on this chip.  The measure-zero
    // case wvno == xk? takes the evanescent arithmetic (p = 0, exp(-0) = 1 gives cos = 1 exactly)
    // and has w/x resp. y/z overridden by selects, which are the values of surfdisp96.f:938-940.
    double cosp, cosq, w, x, y, z;
    double pex = 0.0, sex = 0.0;
    if (wvno < xka) {
        double sinp;
        bh_sincos(p, &sinp, &cosp, LT);
        w = sinp / ra;
        x = -ra * sinp;
    } else {
        pex = p;
        const double fac = (p < 16.0) ? bh_exp((p < 16.0) ? -2.0 * p : -32.0, LT) : 0.0;
        cosp = (1.0 + fac) * 0.5;
        const double sinp = (1.0 - fac) * 0.5;
        const bool eq = (wvno == xka);
        w = eq ? dpth : sinp / ra;
        x = eq ? 0.0 : ra * sinp;
        cosp = eq ? 1.0 : cosp;
        pex = eq ? 0.0 : pex;
    }
    if (wvno < xkb) {
        double sinq;
        bh_sincos(q, &sinq, &cosq, LT);
        y = sinq / rb;
        z = -rb * sinq;
    } else {
        sex = q;
        const double fac = (q < 16.0) ? bh_exp((q < 16.0) ? -2.0 * q : -32.0, LT) : 0.0;
        cosq = (1.0 + fac) * 0.5;
        const double sinq = (1.0 - fac) * 0.5;
        const bool eq = (wvno == xkb);
        y = eq ? dpth : sinq / rb;
        z = eq ? 0.0 : rb * sinq;
        cosq = eq ? 1.0 : cosq;
        sex = eq ? 0.0 : sex;
    }
    const double exa = pex + sex;
    const double a0 = (exa < 60.0) ? bh_exp((exa < 60.0) ? -exa : -60.0, LT) : 0.0;
    o.a0 = a0;
    o.cpcq = cosp * cosq;
    o.cpy = cosp * y;
    o.cpz = cosp * z;
    o.cqw = cosq * w;
    o.cqx = cosq * x;
    o.xy = x * y;
    o.xz = x * z;
    o.wy = w * y;
    o.wz = w * z;
    o.w = w;
    o.cosp = cosp;
}

// The 19 distinct entries of the 5x5 Dunkin compound matrix CA of one layer
// (surfdisp96.f:1024-1068, `dnka`), formed with the reference's operation order.  Stored as
//   c[0..4]  = ca11 ca12 ca13 ca14 ca15          (ca55 = ca11, ca45 = ca12, ca25 = ca14)
//   c[5..7]  = ca21 ca23 ca24                    (ca54 = ca21), ca22 = ca44 = c[8]
//   c[8]     = ca22 (= cpcq)
//   c[9..11] = ca41 ca42 ca43                    (ca52 = ca41)
//   c[12..13]= ca51 ca53
//   c[14..18]= ca31 ca32 ca33 ca34 ca35
struct Ca19 {
    double c[19];
};

__device__ __forceinline__ void rayleigh_ca19(Ca19 &o, double wvno2, double gam, double gammk,
                                              double rho, const LayerTerms &v)
{
    const double two = 2.0;
    const double gamm1 = gam - 1.0;
    const double twgm1 = gam + gamm1;
    const double gmgmk = gam * gammk;
    const double gmgm1 = gam * gamm1;
    const double gm1sq = gamm1 * gamm1;
    const double rho2 = rho * rho;
    const double a0pq = v.a0 - v.cpcq;
    const double ca11 = v.cpcq - two * gmgm1 * a0pq - gmgmk * v.xz - wvno2 * gm1sq * v.wy;
    const double ca12 = (wvno2 * v.cpy - v.cqx) / rho;
    const double ca13 = -(twgm1 * a0pq + gammk * v.xz + wvno2 * gamm1 * v.wy) / rho;
    const double ca14 = (v.cpz - wvno2 * v.cqw) / rho;
    const double ca15 = -(two * wvno2 * a0pq + v.xz + wvno2 * wvno2 * v.wy) / rho2;
    const double ca21 = (gmgmk * v.cpz - gm1sq * v.cqw) * rho;
    const double ca22 = v.cpcq;
    const double ca23 = gammk * v.cpz - gamm1 * v.cqw;
    const double ca24 = -v.wz;
    const double ca41 = (gm1sq * v.cpy - gmgmk * v.cqx) * rho;
    const double ca42 = -v.xy;
    const double ca43 = gamm1 * v.cpy - gammk * v.cqx;
    const double ca51 =
        -(two * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * v.xz + gm1sq * gm1sq * v.wy) * rho2;
    const double ca53 =
        -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * v.xz + gamm1 * gm1sq * v.wy) * rho;
    const double t = -two * wvno2;
    o.c[0] = ca11; o.c[1] = ca12; o.c[2] = ca13; o.c[3] = ca14; o.c[4] = ca15;
    o.c[5] = ca21; o.c[6] = ca23; o.c[7] = ca24; o.c[8] = ca22;
    o.c[9] = ca41; o.c[10] = ca42; o.c[11] = ca43;
    o.c[12] = ca51; o.c[13] = ca53;
    o.c[14] = t * ca53;
    o.c[15] = t * ca43;
    o.c[16] = v.a0 + two * (v.cpcq - ca11);
    o.c[17] = t * ca23;
    o.c[18] = t * ca13;
}

// normc (surfdisp96.f:995-1020): divide the 5-vector by its max-norm (floor 1e-40); the log()
// the Fortran takes of the norm is never used.  max is order-independent, so a tree is used.
// EXACT = false: the five divisions share one refined reciprocal and report their range.
template <bool EXACT>
__device__ __forceinline__ void normalize5(const double ee0, const double ee1, const double ee2,
                                           const double ee3, const double ee4, double e[5],
                                           DivRange &dr)
{
    const double a0 = fabs(ee0), a1 = fabs(ee1), a2 = fabs(ee2), a3 = fabs(ee3), a4 = fabs(ee4);
    double t1 = fmax(fmax(fmax(a0, a1), fmax(a2, a3)), a4);
    if (t1 < 1.0e-40) t1 = 1.0;
    if (EXACT) {
        e[0] = ee0 / t1;
        e[1] = ee1 / t1;
        e[2] = ee2 / t1;
        e[3] = ee3 / t1;
        e[4] = ee4 / t1;
    } else {
        dr.see(fmin(fmin(fmin(a0, a1), fmin(a2, a3)), a4));
        dr.see(t1);
        const double r = bh_rcp_refined(t1);
        e[0] = bh_quot(ee0, t1, r);
        e[1] = bh_quot(ee1, t1, r);
        e[2] = bh_quot(ee2, t1, r);
        e[3] = bh_quot(ee3, t1, r);
        e[4] = bh_quot(ee4, t1, r);
    }
}

// e <- normalise(e * CA): ee(i) = sum_j e(j)*ca(j,i) accumulated from 0.0 in j order
// (surfdisp96.f:836-842), then normc (:995-1020; its log() result is never used).
template <bool EXACT>
__device__ __forceinline__ void rayleigh_apply(double e[5], const double *c, DivRange &dr)
{
    const double ca11 = c[0], ca12 = c[1], ca13 = c[2], ca14 = c[3], ca15 = c[4];
    const double ca21 = c[5], ca23 = c[6], ca24 = c[7], ca22 = c[8];
    const double ca41 = c[9], ca42 = c[10], ca43 = c[11], ca51 = c[12], ca53 = c[13];
    const double ca31 = c[14], ca32 = c[15], ca33 = c[16], ca34 = c[17], ca35 = c[18];
    const double ca25 = ca14, ca44 = ca22, ca45 = ca12, ca52 = ca41, ca54 = ca21, ca55 = ca11;
    double ee0 = 0.0, ee1 = 0.0, ee2 = 0.0, ee3 = 0.0, ee4 = 0.0;
    ee0 = ee0 + e[0] * ca11; ee0 = ee0 + e[1] * ca21; ee0 = ee0 + e[2] * ca31; ee0 = ee0 + e[3] * ca41; ee0 = ee0 + e[4] * ca51;
    ee1 = ee1 + e[0] * ca12; ee1 = ee1 + e[1] * ca22; ee1 = ee1 + e[2] * ca32; ee1 = ee1 + e[3] * ca42; ee1 = ee1 + e[4] * ca52;
    ee2 = ee2 + e[0] * ca13; ee2 = ee2 + e[1] * ca23; ee2 = ee2 + e[2] * ca33; ee2 = ee2 + e[3] * ca43; ee2 = ee2 + e[4] * ca53;
    ee3 = ee3 + e[0] * ca14; ee3 = ee3 + e[1] * ca24; ee3 = ee3 + e[2] * ca34; ee3 = ee3 + e[3] * ca44; ee3 = ee3 + e[4] * ca54;
    ee4 = ee4 + e[0] * ca15; ee4 = ee4 + e[1] * ca25; ee4 = ee4 + e[2] * ca35; ee4 = ee4 + e[3] * ca45; ee4 = ee4 + e[4] * ca55;
    normalize5<EXACT>(ee0, ee1, ee2, ee3, ee4, e, dr);
}

template <bool EXACT>
__device__ __forceinline__ void rayleigh_layer(double e[5], double wvno2, double gam, double gammk,
                                               double rho, const LayerTerms &v, DivRange &dr)
{
    Ca19 ca;
    rayleigh_ca19(ca, wvno2, gam, gammk, rho, v);
    rayleigh_apply<EXACT>(e, ca.c, dr);
}

// ---- Rayleigh: Dunkin compound-matrix secular function (surfdisp96.f:773-871) -----------------
template <bool EXACT>
__device__ double rayleigh_secular(double wvno, double omga, const ModelLds &md, int mmax, int llw,
                                   int mtop, DivRange &dr, const LibmTabs &LT)
{
    double e[5];
    LayerTerms v;
    double omega = omga;
    if (omega < 1.0e-4) omega = 1.0e-4;
    const double wvno2 = wvno * wvno;
    {
        const double ah = md.A(mmax - 1), bh = md.Bv(mmax - 1);
        const double xka = omega / ah;
        const double xkb = omega / bh;
        double wvnop = wvno + xka;
        double wvnom = fabs(wvno - xka);
        const double ra = sqrt(wvnop * wvnom);
        wvnop = wvno + xkb;
        wvnom = fabs(wvno - xkb);
        const double rb = sqrt(wvnop * wvnom);
        const double t = bh / omega;
        const double gammk = 2.0 * t * t;
        const double gam = gammk * wvno2;
        const double gamm1 = gam - 1.0;
        const double rho1 = md.R(mmax - 1);
        e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
        e[1] = -rho1 * ra;
        e[2] = rho1 * (gamm1 - gammk * ra * rb);
        e[3] = rho1 * rb;
        e[4] = wvno2 - ra * rb;
    }
    for (int m = mtop - 2; m >= 0; --m) {
        if (m <= mmax - 2 && m >= llw - 1) {
            const double am = md.A(m), bm = md.Bv(m);
            const double xka = omega / am;
            const double xkb = omega / bm;
            const double t = bm / omega;
            const double gammk = 2.0 * t * t;
            const double gam = gammk * wvno2;
            double wvnop = wvno + xka;
            double wvnom = fabs(wvno - xka);
            const double ra = sqrt(wvnop * wvnom);
            wvnop = wvno + xkb;
            wvnom = fabs(wvno - xkb);
            const double rb = sqrt(wvnop * wvnom);
            const double dpth = md.D(m);
            const double rho1 = md.R(m);
            const double p = ra * dpth;
            const double q = rb * dpth;
            layer_products(p, q, ra, rb, wvno, xka, xkb, dpth, v, LT);
            rayleigh_layer<EXACT>(e, wvno2, gam, gammk, rho1, v, dr);
        }
    }
    double result = e[0];
    if (llw != 1) { // water layer on top (surfdisp96.f:850-866); unreachable from BayHunter
        const double xka = omega / md.A(0);
        const double wvnop = wvno + xka;
        const double wvnom = fabs(wvno - xka);
        const double ra = sqrt(wvnop * wvnom);
        const double dpth = md.D(0);
        const double rho1 = md.R(0);
        const double p = ra * dpth;
        const double znul = 1.0e-5;
        layer_products(p, znul, ra, znul, wvno, xka, znul, dpth, v, LT);
        const double w0 = -rho1 * v.w;
        result = v.cosp * e[0] + w0 * e[1];
    }
    return result;
}

// ---- half-space Rayleigh velocity, binary32 throughout (surfdisp96.f:367-388) -----------------
__device__ float gtsolh_f32(float a, float b)
{
    float c = 0.95f * b;
    for (int i = 0; i < 5; ++i) {
        const float gamma = b / a;
        const float kappa = c / b;
        const float k2 = kappa * kappa;
        const float gk = gamma * kappa;
        const float gk2 = gk * gk;
        const float fac1 = sqrtf(1.0f - gk2);
        const float fac2 = sqrtf(1.0f - k2);
        const float tk = 2.0f - k2;
        const float fr = tk * tk - 4.0f * fac1 * fac2;
        float frp = -4.0f * (2.0f - k2) * kappa + 4.0f * fac2 * gamma * gamma * kappa / fac1 +
                    4.0f * fac1 * kappa / fac2;
        frp = frp / b;
        c = c - fr / frp;
    }
    return c;
}

// continuation tags: what the pending secular evaluation is for
enum : int {
    ST_FIRST = 0, // del1 at the start value c1                      (surfdisp96.f:421-423)
    ST_STEP = 1,  // del2 at c2 = c1 +- dc                           (:447-449)
    ST_NEV0 = 2,  // first midpoint inside nevill                    (:582-583)
    ST_NEVL = 3,  // midpoint / Neville estimate, then top of loop   (:586-...)
    ST_NEVF = 4   // forced midpoint after the estimate left the bracket (:594-598)
};

// ---- the per-model search state machine -------------------------------------------------------
// Everything the reference's driver (surfdisp96.f:172-357), getsol (:390-482) and nevill
// (:557-686) keep between two secular-function evaluations, for the fundamental mode.
// `advance(del)` consumes the value of the secular function at `ceval` and either finishes the
// model or leaves the next phase velocity to evaluate in `ceval` (with `omega` current).
template <int XSC> // XSC > 0: compile-time lane stride of the Neville tables in LDS; 0: run-time (member XS)
struct SearchT {
    int XS = XSC;
    // constants of the reference's driver (compile-time: they cost no registers)
    static constexpr double one = 1.0e-2;
    static constexpr double onea = (double)1.5f;
    static constexpr double dc = (double)0.005f;     // abs(dble(0.005)) with a default-real literal
    static constexpr double twopi = 2.0 * 3.141592653589793;
    static constexpr double pct = (double)0.01f;     // `0.01*ss1` with a default-real literal (:623-626)
    double cm, betmxd;
    bool group;
    int K;
    int mode;            // highest mode wanted (1 = fundamental)
    double *cper, *cbper; // LDS, only for mode > 1: c(k) / cb(k) of surfdisp96.f:85, element k at [k*XS]
    const double *per; // LDS
    double *xl, *yl;   // LDS Neville tables, element j at [j*XS]
    double *vel;       // this model's output row (global)
    bool writer;       // this lane stores results (one lane per model)
    // state
    int k, root, st, ifirst, idir, nev, mnev, nctrl, errflag, iq, ift;
    bool active;
    double c1, c2, clow, del1, del2, del1st, c3, del3, ck, t1, omega, ceval;
    float t1a, t1b;
    unsigned int evals;

    __device__ __forceinline__ void set_period(int kk)
    {
        const float h32 = 0.005f;
        double tt = per[kk];
        if (group) {
            t1a = (float)(tt / (double)(1.0f + h32));
            t1b = (float)(tt / (double)(1.0f - h32));
            tt = (double)t1a;
        } else {
            t1a = (float)tt;
        }
        t1 = tt;
        omega = twopi / t1;
    }

    // driver set-up (surfdisp96.f:124-217): extremal velocities, start value
    template <class MD>
    __device__ void init(const MD &md, int mmax, bool valid, int igr, int K_, const double *per_,
                         double *xl_, double *yl_, double *vel_, bool writer_, int mode_ = 1,
                         double *cper_ = nullptr, double *cbper_ = nullptr)
    {
        float betmx = -1.e20f, betmn = 1.e20f;
        int jmn = 0, jsol = 1;
        // Input sanity.  The reference's loops are bounded only through the model's velocities: with a NaN
        // or an absurd value in the model it walks the velocity axis (practically) for ever.  A GPU kernel
        // must end: such a model is reported in-band as failed (err = 1, zeros) without being searched.
        bool sane = true;
        for (int i = 0; i < mmax; ++i) {
            const float bi = md.Bf(i), ai = md.Af(i);
            const float di = md.Df(i), ri = (float)md.R(i);
            sane = sane && (ai > 0.0f) && (ai <= 100.0f) && (bi >= 0.0f) && (bi <= 100.0f) && (ri > 0.0f) && (ri < 1.0e6f) &&
                   (i == mmax - 1 || (di >= 0.0f && di < 1.0e7f));
            if (bi > 0.01f && bi < betmn) {
                betmn = bi;
                jmn = i;
                jsol = 1;
            } else if (bi <= 0.01f && ai < betmn) {
                betmn = ai;
                jmn = i;
                jsol = 0;
            }
            if (bi > betmx) betmx = bi;
        }
        float cc1 = (jsol == 0) ? betmn : gtsolh_f32(md.Af(jmn), md.Bf(jmn));
        cc1 = 0.95f * cc1;
        cc1 = 0.90f * cc1;
        const double cc = (double)cc1;
        cm = cc;
        betmxd = (double)betmx;
        group = igr > 0;
        K = K_;
        per = per_;
        xl = xl_;
        yl = yl_;
        vel = vel_;
        writer = writer_;
        mode = mode_;
        cper = cper_;
        cbper = cbper_;
        if (mode > 1)
            for (int i = 0; i < K; ++i) { // do 450: c() = cb() = 0 (every lane of the group writes the same)
                cper[i * XS] = 0.0;
                cbper[i * XS] = 0.0;
            }
        iq = 1;
        ift = 999;
        k = 0; root = 0; st = ST_FIRST; ifirst = 1;
        active = valid && K > 0 && sane;
        errflag = 0;
        if (valid && !sane) {
            errflag = 1;
            if (writer_)
                for (int i = 0; i < K_; ++i) vel_[i] = 0.0;
        }
        c1 = cc; c2 = 0.0; clow = cc; del1 = del2 = del1st = 0.0;
        c3 = del3 = ck = 0.0;
        idir = 1; nev = 1; mnev = 1; nctrl = 1;
        t1a = t1b = 0.f;
        t1 = 1.0; omega = 1.0;
        evals = 0;
        if (active) set_period(0);
        ceval = c1;
    }

    // label 1700/1750: the current mode found no root at period k
    __device__ __forceinline__ void fail_mode()
    {
        if (iq == 1) errflag = 1; // higher modes fail silently (:313)
        ift = k;
        if (writer)
            for (int i = k; i < K; ++i) vel[i] = 0.0;
    }

    // Set up the root search of period k of mode iq (initial guess logic, :253-272), moving on to
    // the next mode when the period list is exhausted or a previous mode already failed here.
    __device__ void next_search()
    {
        for (;;) {
            bool over = (k >= K);
            if (!over && k >= ift) { // `if(k.ge.ift) go to 1700`
                fail_mode();
                over = true;
            }
            if (!over) break;
            if (iq >= mode) {
                active = false;
                return;
            }
            iq = iq + 1;
            k = 0;
        }
        set_period(k);
        root = 0;
        if (mode == 1) { // fundamental mode only: c(k-1) is still in a register
            ifirst = 0;
            c1 = ck - onea * dc;
            clow = cm;
        } else if (k == 0) {
            c1 = cper[0] + one * dc; // iq > 1 here (iq == 1, k == 0 is set up by init)
            clow = c1;
            ifirst = 1;
        } else if (iq > 1) {
            ifirst = 0;
            clow = cper[k * XS] + one * dc;
            c1 = cper[(k - 1) * XS];
            if (c1 < clow) c1 = clow;
        } else {
            ifirst = 0;
            c1 = cper[(k - 1) * XS] - onea * dc;
            clow = cm;
        }
        st = ST_FIRST;
        ceval = c1;
    }

    // Look-ahead: candidate 0 is the pending request; candidate r > 0 is the phase velocity the r-th
    // request from now will most probably be for -- further bracket steps while stepping (:437-449),
    // further halvings towards the side on which a straight line through the bracket ends puts the
    // root while refining (:600-660).  Purely a guess about which values will be asked for: a value
    // is only ever consumed by advance() if it was computed for exactly the (ceval, omega) requested.
    __device__ __forceinline__ double candidate(int r) const
    {
        double q = ceval;
        if (st == ST_FIRST || st == ST_STEP) {
            const bool up = (st == ST_FIRST) || (idir > 0);
            for (int j = 0; j < r; ++j) q = up ? q + dc : q - dc;
        } else {
            double lo = c1, hi = c2; // the function keeps the sign of del1 at `lo`
            const double w = c2 - c1;
            for (int j = 0; j < r; ++j) {
                const double t = del1 * (c2 - q) + del2 * (q - c1); // (c2 - c1) * linear model at q
                const bool neg_lin = (t < 0.0) != (w < 0.0);
                const bool differs = neg_lin != (del1 < 0.0);
                lo = differs ? lo : q;
                hi = differs ? q : hi;
                q = 0.5 * (lo + hi);
            }
        }
        return q;
    }

    __device__ void advance(double del)
    {
        ++evals;
        // `todo`: 0 nothing, 1 prepare next bracket step, 2 root search failed (iret = -1),
        // 3 refinement finished with c3, 4 nevill top-of-loop, 5 nevill post-bracket section,
        // 6 root found
        int todo = 0;
        switch (st) {
        case ST_FIRST:
            del1 = del;
            if (ifirst == 1) del1st = del1;
            idir = (ifirst != 1 && signs_differ(del1st, del1)) ? -1 : +1;
            todo = 1;
            break;
        case ST_STEP:
            del2 = del;
            if (signs_differ(del1, del2)) { // bracketed: enter nevill with (c1,c2,del1,del2)
                c3 = 0.5 * (c1 + c2);
                ceval = c3;
                st = ST_NEV0;
            } else {
                c1 = c2;
                del1 = del2;
                if (c1 < cm || c1 >= betmxd + dc) todo = 2;
                else todo = 1;
            }
            break;
        case ST_NEV0:
            del3 = del;
            nev = 1;
            nctrl = 1;
            mnev = 1;
            todo = 4;
            break;
        case ST_NEVL:
            del3 = del;
            todo = 4;
            break;
        case ST_NEVF:
            del3 = del;
            todo = 5;
            break;
        }
        if (todo == 4) { // label 100 of nevill
            nctrl = nctrl + 1;
            if (nctrl >= 100) {
                todo = 3;
            } else if (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)) {
                nev = 0;
                c3 = 0.5 * (c1 + c2);
                ceval = c3;
                st = ST_NEVF;
                todo = 0;
            } else {
                todo = 5;
            }
        }
        if (todo == 5) {
            const double s13 = del1 - del3;
            const double s32 = del3 - del2;
            if (signs_differ(del3, del1)) {
                c2 = c3;
                del2 = del3;
            } else {
                c1 = c3;
                del1 = del3;
            }
            if (fabs(c1 - c2) <= 1.0e-6 * c1) {
                todo = 3;
            } else {
                if (signs_differ(s13, s32)) nev = 0;
                const double ss1 = fabs(del1), s1 = pct * ss1;
                const double ss2 = fabs(del2), s2 = pct * ss2;
                bool halve = (s1 > ss2 || s2 > ss1 || nev == 0);
                if (!halve) {
                    if (nev == 2) {
                        xl[mnev * XS] = c3;
                        yl[mnev * XS] = del3;
                    } else {
                        xl[0] = c1;
                        yl[0] = del1;
                        xl[XS] = c2;
                        yl[XS] = del2;
                        mnev = 1;
                    }
                    const double ym = yl[mnev * XS];
                    for (int kk = 1; kk <= mnev; ++kk) {
                        const int j = mnev - kk;
                        const double yj = yl[j * XS];
                        const double denom = ym - yj;
                        if (fabs(denom) < 1.0e-10 * fabs(ym)) {
                            halve = true;
                            break;
                        }
                        xl[j * XS] = (-yj * xl[(j + 1) * XS] + ym * xl[j * XS]) / denom;
                    }
                    if (!halve) {
                        c3 = xl[0];
                        nev = 2;
                        mnev = mnev + 1;
                        if (mnev > 10) mnev = 10;
                    }
                }
                if (halve) {
                    c3 = 0.5 * (c1 + c2);
                    nev = 1;
                    mnev = 1;
                }
                ceval = c3;
                st = ST_NEVL;
                todo = 0;
            }
        }
        if (todo == 3) { // getsol after nevill (:468-471)
            c1 = c3;
            todo = (c1 > betmxd) ? 2 : 6;
        }
        if (todo == 2 || todo == 6) { // a root search ended: 6 = found c1, 2 = failed
            bool period_done = false;
            double c1b = 0.0; // the "c1" the driver uses after the (optional) second search
            if (root == 0) {
                if (todo == 2) { // no root: err (fundamental mode only), zero-fill, next mode (:313-354)
                    fail_mode();
                    if (iq >= mode) {
                        active = false;
                    } else {
                        iq = iq + 1;
                        k = 0;
                        next_search();
                    }
                } else {
                    ck = c1;
                    if (mode > 1) cper[k * XS] = c1;
                    if (group) { // second root at the slightly longer period (:282-287)
                        root = 1;
                        t1 = (double)t1b;
                        omega = twopi / t1;
                        ifirst = 0;
                        clow = ((mode > 1) ? cbper[k * XS] : 0.0) + one * dc; // cb(k) of the previous mode
                        c1 = c1 - onea * dc;
                        st = ST_FIRST;
                        ceval = c1;
                    } else {
                        period_done = true;
                    }
                }
            } else {
                c1b = (todo == 2) ? ck : c1; // second root failed: reuse the first (:291-293)
                if (mode > 1) cbper[k * XS] = c1b;
                period_done = true;
            }
            if (period_done) {
                const float cc0 = (float)ck;
                double out;
                if (!group) {
                    out = (double)cc0;
                } else { // all binary32 (:305)
                    const float cc1s = (float)c1b;
                    const float gvel =
                        (1.0f / t1a - 1.0f / t1b) / (1.0f / (t1a * cc0) - 1.0f / (t1b * cc1s));
                    out = (double)gvel;
                }
                if (writer) vel[k] = out;
                k = k + 1;
                next_search();
            }
            todo = 0;
        }
        if (todo == 1) { // label 1000 of getsol: next bracket step (:437-446)
            c2 = (idir > 0) ? c1 + dc : c1 - dc;
            if (c2 <= clow) {
                idir = +1;
                c1 = clow;
                c2 = c1 + dc;
                // dc > 0, so the retried c2 = clow + dc is above clow: no further loop
            }
            ceval = c2;
            st = ST_STEP;
        }
    }

    // ---- advance() for SIMT execution ---------------------------------------------------------------
    // The same transition as advance(), written as straight-line selects: in a wavefront whose models are
    // at different points of their searches every `if` of advance() is an exec-mask region that all lanes
    // walk through (measured: ~2000 cycles per transition); here only the two rare, expensive steps stay
    // behind a branch -- the Neville interpolation (divisions, LDS table) and the end of a root search
    // (result store, set-up of the next period).  Every value is produced by the operations of advance().
    __device__ __forceinline__ void advance2(double del)
    {
        ++evals;
        const bool sF = st == ST_FIRST, sS = st == ST_STEP, sN0 = st == ST_NEV0, sNF = st == ST_NEVF;
        const bool isN = st >= ST_NEV0;
        // -- first value of a search (:421-423) / a bracket step (:447-449)
        double n_del1 = sF ? del : del1;
        del1st = (sF && ifirst == 1) ? del : del1st;
        int n_idir = sF ? ((ifirst != 1 && signs_differ(del1st, n_del1)) ? -1 : +1) : idir;
        del2 = sS ? del : del2;
        const bool brk = sS && signs_differ(del1, del);   // bracketed: enter nevill with (c1,c2,del1,del2)
        const bool nb = sS && !brk;
        double n_c1 = nb ? c2 : c1;
        n_del1 = nb ? del : n_del1;
        const bool failS = nb && (n_c1 < cm || n_c1 >= betmxd + dc);
        const bool step = sF || (nb && !failS);             // label 1000 of getsol: next bracket step (:437-446)
        double c2n = (n_idir > 0) ? n_c1 + dc : n_c1 - dc;
        const bool redir = step && c2n <= clow;
        n_idir = redir ? +1 : n_idir;
        n_c1 = redir ? clow : n_c1;
        c2n = redir ? clow + dc : c2n;
        // -- inside nevill (:582-660)
        del3 = isN ? del : del3;
        int n_nev = sN0 ? 1 : nev, n_mnev = sN0 ? 1 : mnev;
        const bool t4 = isN && !sNF;                          // label 100
        int n_nctrl = sN0 ? 1 : nctrl;
        n_nctrl = t4 ? n_nctrl + 1 : n_nctrl;
        const bool fin100 = t4 && n_nctrl >= 100;
        const bool outside = t4 && !fin100 && (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)); // estimate left the bracket
        const bool t5 = sNF || (t4 && !fin100 && !outside);
        const double s13 = del1 - del3, s32 = del3 - del2;
        const bool sd31 = signs_differ(del3, del1);
        const bool upd2 = t5 && sd31, upd1 = t5 && !sd31;
        const double b_c2 = upd2 ? c3 : c2, b_del2 = upd2 ? del3 : del2;
        const double b_c1 = upd1 ? c3 : n_c1, b_del1 = upd1 ? del3 : n_del1;
        const bool conv = t5 && (fabs(b_c1 - b_c2) <= 1.0e-6 * b_c1);
        const bool t5c = t5 && !conv;
        n_nev = (outside || (t5c && signs_differ(s13, s32))) ? 0 : n_nev;
        const double ss1 = fabs(b_del1), s1 = pct * ss1;
        const double ss2 = fabs(b_del2), s2 = pct * ss2;
        bool halve = (s1 > ss2 || s2 > ss1 || n_nev == 0);
        double c3n = c3;
        if (t5c && !halve) { // inverse Neville interpolation (:626-655)
            if (n_nev == 2) {
                xl[n_mnev * XS] = c3;
                yl[n_mnev * XS] = del3;
            } else {
                xl[0] = b_c1;
                yl[0] = b_del1;
                xl[XS] = b_c2;
                yl[XS] = b_del2;
                n_mnev = 1;
            }
            const double ym = yl[n_mnev * XS];
            for (int kk = 1; kk <= n_mnev; ++kk) {
                const int j = n_mnev - kk;
                const double yj = yl[j * XS];
                const double denom = ym - yj;
                if (fabs(denom) < 1.0e-10 * fabs(ym)) {
                    halve = true;
                    break;
                }
                xl[j * XS] = (-yj * xl[(j + 1) * XS] + ym * xl[j * XS]) / denom;
            }
            if (!halve) {
                c3n = xl[0];
                n_nev = 2;
                n_mnev = n_mnev + 1;
                if (n_mnev > 10) n_mnev = 10;
            }
        }
        const bool mid = brk || outside || (t5c && halve);  // the next point is the middle of the bracket
        const double midc = 0.5 * (b_c1 + b_c2);
        c3n = mid ? midc : c3n;
        n_nev = (t5c && halve) ? 1 : n_nev;
        n_mnev = (t5c && halve) ? 1 : n_mnev;
        // -- commit
        c1 = b_c1; del1 = b_del1;
        c2 = step ? c2n : b_c2;
        del2 = b_del2;
        idir = n_idir; nev = n_nev; mnev = n_mnev; nctrl = n_nctrl;
        const bool refine = brk || outside || t5c;
        c3 = refine ? c3n : c3;
        ceval = step ? c2n : (refine ? c3n : ceval);
        st = step ? (int)ST_STEP : (brk ? (int)ST_NEV0 : (outside ? (int)ST_NEVF : (t5c ? (int)ST_NEVL : st)));
        const bool fin = fin100 || conv;                      // getsol after nevill (:468-471)
        c1 = fin ? c3 : c1;
        const bool ended = fin || failS;
        if (ended) end_of_search((fin && !(c1 > betmxd)) ? 6 : 2);
    }

    // A root search ended: 6 = found c1, 2 = failed.  (The driver's part, surfdisp96.f:276-354.)
    __device__ void end_of_search(int todo)
    {
        bool period_done = false;
        double c1b = 0.0; // the "c1" the driver uses after the (optional) second search
        if (root == 0) {
            if (todo == 2) { // no root: err (fundamental mode only), zero-fill, next mode (:313-354)
                fail_mode();
                if (iq >= mode) {
                    active = false;
                } else {
                    iq = iq + 1;
                    k = 0;
                    next_search();
                }
            } else {
                ck = c1;
                if (mode > 1) cper[k * XS] = c1;
                if (group) { // second root at the slightly longer period (:282-287)
                    root = 1;
                    t1 = (double)t1b;
                    omega = twopi / t1;
                    ifirst = 0;
                    clow = ((mode > 1) ? cbper[k * XS] : 0.0) + one * dc; // cb(k) of the previous mode
                    c1 = c1 - onea * dc;
                    st = ST_FIRST;
                    ceval = c1;
                } else {
                    period_done = true;
                }
            }
        } else {
            c1b = (todo == 2) ? ck : c1; // second root failed: reuse the first (:291-293)
            if (mode > 1) cbper[k * XS] = c1b;
            period_done = true;
        }
        if (period_done) {
            const float cc0 = (float)ck;
            double out;
            if (!group) {
                out = (double)cc0;
            } else { // all binary32 (:305)
                const float cc1s = (float)c1b;
                const float gvel =
                    (1.0f / t1a - 1.0f / t1b) / (1.0f / (t1a * cc0) - 1.0f / (t1b * cc1s));
                out = (double)gvel;
            }
            if (writer) vel[k] = out;
            k = k + 1;
            next_search();
        }
    }

    // The trial velocities of the next round, candidate(0..n-1), computed incrementally; lane keeps those
    // with index ia / ib.  Same operations as candidate().
    __device__ __forceinline__ void candidates(int n, int ia, int ib, double &qa, double &qb) const
    {
        const bool stepping = (st == ST_FIRST || st == ST_STEP);
        const bool up = (st == ST_FIRST) || (idir > 0);
        double q = ceval, lo = c1, hi = c2;
        const double w = c2 - c1;
        qa = q;
        qb = q;
        for (int j = 1; j < n; ++j) {
            const double qs = up ? q + dc : q - dc;
            const double t = del1 * (c2 - q) + del2 * (q - c1); // (c2 - c1) * linear model at q
            const bool neg_lin = (t < 0.0) != (w < 0.0);
            const bool differs = neg_lin != (del1 < 0.0);
            lo = differs ? lo : q;
            hi = differs ? q : hi;
            const double qr = 0.5 * (lo + hi);
            q = stepping ? qs : qr;
            qa = (j == ia) ? q : qa;
            qb = (j == ib) ? q : qb;
        }
    }

    // candidate(0..n-1) written to q[0..n-1] (LDS): the requests of the next round.
    __device__ __forceinline__ void candidates_store(int n, double *qout) const
    {
        const bool stepping = (st == ST_FIRST || st == ST_STEP);
        const bool up = (st == ST_FIRST) || (idir > 0);
        double q = ceval, lo = c1, hi = c2;
        const double w = c2 - c1;
        qout[0] = q;
        for (int j = 1; j < n; ++j) {
            const double qs = up ? q + dc : q - dc;
            const double t = del1 * (c2 - q) + del2 * (q - c1); // (c2 - c1) * linear model at q
            const bool neg_lin = (t < 0.0) != (w < 0.0);
            const bool differs = neg_lin != (del1 < 0.0);
            lo = differs ? lo : q;
            hi = differs ? q : hi;
            const double qr = 0.5 * (lo + hi);
            q = stepping ? qs : qr;
            qout[j] = q;
        }
    }

    // ---- parking: everything advance() changes, in ST_SLOTS doubles (the group kernel keeps the
    // search state of a model in LDS between two state transitions, not in registers) ----------
    static constexpr int ST_SLOTS = 18;
    __device__ __forceinline__ void park(double *p) const
    {
        const unsigned w0 = (unsigned)st | ((unsigned)ifirst << 3) | ((idir > 0 ? 1u : 0u) << 4) | ((unsigned)nev << 5) |
                            ((unsigned)mnev << 7) | ((unsigned)root << 11) | ((active ? 1u : 0u) << 12) |
                            ((unsigned)errflag << 13) | ((unsigned)iq << 14);
        const unsigned w1 = (unsigned)k | ((unsigned)nctrl << 8) | ((unsigned)ift << 16);
        double2 *q = reinterpret_cast<double2 *>(p);
        q[0] = make_double2(c1, c2);
        q[1] = make_double2(del1, del2);
        q[2] = make_double2(c3, del3);
        q[3] = make_double2(clow, del1st);
        q[4] = make_double2(ck, t1);
        q[5] = make_double2(omega, ceval);
        q[6] = make_double2(cm, betmxd);
        q[7] = make_double2(__hiloint2double((int)__float_as_uint(t1b), (int)__float_as_uint(t1a)),
                            __hiloint2double((int)w1, (int)w0));
        q[8] = make_double2(__hiloint2double(0, (int)evals), 0.0);
    }
    __device__ __forceinline__ void unpark(const double *p)
    {
        const double2 *q = reinterpret_cast<const double2 *>(p);
        const double2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4], a5 = q[5], a6 = q[6], a7 = q[7];
        c1 = a0.x; c2 = a0.y; del1 = a1.x; del2 = a1.y; c3 = a2.x; del3 = a2.y; clow = a3.x; del1st = a3.y;
        ck = a4.x; t1 = a4.y; omega = a5.x; ceval = a5.y; cm = a6.x; betmxd = a6.y;
        t1a = __uint_as_float((unsigned)__double2loint(a7.x));
        t1b = __uint_as_float((unsigned)__double2hiint(a7.x));
        const unsigned w0 = (unsigned)__double2loint(a7.y), w1 = (unsigned)__double2hiint(a7.y);
        st = (int)(w0 & 7u); ifirst = (int)((w0 >> 3) & 1u); idir = ((w0 >> 4) & 1u) ? 1 : -1;
        nev = (int)((w0 >> 5) & 3u); mnev = (int)((w0 >> 7) & 15u); root = (int)((w0 >> 11) & 1u);
        active = ((w0 >> 12) & 1u) != 0u; errflag = (int)((w0 >> 13) & 1u); iq = (int)((w0 >> 14) & 31u);
        k = (int)(w1 & 255u); nctrl = (int)((w1 >> 8) & 255u); ift = (int)(w1 >> 16);
        evals = (unsigned)__double2loint(q[8].x);
    }
};
using SearchRt = SearchT<0>;

// =================================================================================================
// Kernel 1: one lane = one model (G = 1).  Best when the batch alone fills the chip
// (B*targets/64 >> 1024 waves); everything per-layer stays in registers.
// =================================================================================================
template <int IFUNC>
__global__ __launch_bounds__(BH_WAVE) void swd_kernel(SwdKernelArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int sidx = blockIdx.x * BH_WAVE + lane; // position in the processing order
    const bool valid = sidx < A.B;
    const int ib = valid ? (A.perm ? A.perm[sidx] : sidx) : 0;
    const int Lmax = A.Lmax;
    const int K = A.K;

    float *mdl = reinterpret_cast<float *>(smem);                       // [4][Lmax][64]
    double *xs = reinterpret_cast<double *>(smem + (size_t)4 * Lmax * BH_WAVE * sizeof(float));
    double *ys = xs + NEV_MAX * BH_WAVE;                                 // [11][64] each
    double *per = ys + NEV_MAX * BH_WAVE;                                // [K]
    unsigned char *after = reinterpret_cast<unsigned char *>(per + ((K + 1) & ~1));
    const LibmTabs LT = stage_libm_tables(after, lane);
    double *cpl = reinterpret_cast<double *>(after + LIBM_TAB_BYTES); // [2][K][64], only if mode > 1

    for (int k = lane; k < K; k += BH_WAVE) per[k] = A.periods[k];

    // ---- stage the model through LDS, rounding to binary32 like the f2py boundary -----------
    const int mmax = valid ? A.nlay[ib] : 2;
    int mtop = mmax; // wave-wide maximum layer count = loop bound of the secular functions
    for (int off = 32; off > 0; off >>= 1) mtop = max(mtop, __shfl_xor(mtop, off));
    {
        const ptrdiff_t base = (ptrdiff_t)ib * A.sb;
        for (int l = 0; l < Lmax; ++l) {
            float fd = 0.f, fa = 1.f, fb = 1.f, fr = 1.f;
            if (valid && l < mmax) {
                const ptrdiff_t o = base + (ptrdiff_t)l * A.sl;
                fd = (float)A.h[o];
                fa = (float)A.vp[o];
                fb = (float)A.vs[o];
                fr = (float)A.rho[o];
            }
            mdl[(0 * Lmax + l) * BH_WAVE + lane] = fd;
            mdl[(1 * Lmax + l) * BH_WAVE + lane] = fa;
            mdl[(2 * Lmax + l) * BH_WAVE + lane] = fb;
            mdl[(3 * Lmax + l) * BH_WAVE + lane] = fr;
        }
    }
    __syncthreads();
    ModelLds md;
    md.d = mdl + 0 * Lmax * BH_WAVE + lane;
    md.a = mdl + 1 * Lmax * BH_WAVE + lane;
    md.b = mdl + 2 * Lmax * BH_WAVE + lane;
    md.rho = mdl + 3 * Lmax * BH_WAVE + lane;
    const int llw = (md.Bf(0) <= 0.0f) ? 2 : 1;

    SearchT<BH_WAVE> S;
    S.init(md, mmax, valid, A.igr, K, per, xs + lane, ys + lane, A.vel + (size_t)ib * A.ldv, true, A.mode,
           cpl + lane, cpl + (size_t)K * BH_WAVE + lane);

    while (__ballot(S.active) != 0ull) {
        if (!S.active) continue;
        const double wvno = S.omega / S.ceval;
        double del;
        DivRange dr;
        dr.reset();
        if (IFUNC == 1)
            del = love_secular<false>(wvno, S.omega, md, mmax, llw, mtop, dr, LT);
        else
            del = rayleigh_secular<false>(wvno, S.omega, md, mmax, llw, mtop, dr, LT);
        if (!dr.ok()) { // operands left the range the fast divisions are exact in: redo verbatim
            if (IFUNC == 1)
                del = love_secular<true>(wvno, S.omega, md, mmax, llw, mtop, dr, LT);
            else
                del = rayleigh_secular<true>(wvno, S.omega, md, mmax, llw, mtop, dr, LT);
        }
        S.advance(del);
    }
    if (valid) A.err[ib] = S.errflag;
    if (A.neval != nullptr) {
        unsigned long long tot = S.evals;
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        if (lane == 0) atomicAdd(A.neval, tot);
    }
}

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
template <bool PAR5, bool RAGGED, bool EXACT>
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
            const double n0 = cn[0], n1 = cn[1], n2 = cn[2], n3 = cn[3], n4 = cn[4];
            double ee = 0.0;
            ee = ee + e[0] * c0;
            ee = ee + e[1] * c1;
            ee = ee + e[2] * c2;
            ee = ee + e[3] * c3;
            ee = ee + e[4] * c4;
            const double v0 = __shfl(ee, gbase + 0), v1 = __shfl(ee, gbase + 1), v2 = __shfl(ee, gbase + 2),
                         v3 = __shfl(ee, gbase + 3), v4 = __shfl(ee, gbase + 4);
            double en[5];
            DivRange d2 = dr;
            normalize5<EXACT>(v0, v1, v2, v3, v4, en, d2);
            if (on) dr = d2;
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
#pragma unroll
                for (int j = 0; j < 5; ++j) acc = acc + e[j] * cc[5 * i + j];
                ee[i] = acc;
            }
            DivRange d2 = dr;
            normalize5<EXACT>(ee[0], ee[1], ee[2], ee[3], ee[4], en, d2);
            if (on) dr = d2;
#pragma unroll
            for (int i = 0; i < 5; ++i) e[i] = on ? en[i] : e[i];
        }
    }
}

// Phase B of the group kernel, Love: parked per layer (cosq, y, z, xmu, rcp(xmu)).
template <bool RAGGED, bool EXACT>
__device__ __forceinline__ void love_chain_group(double &e1, double &e2, const double *cam, int mtop,
                                                 int mmax, int llw, DivRange &dr)
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
        love_step<EXACT>(n1, n2, p0.x, p0.y, p1.x, p1.y, p2.x, d2);
        if (on) {
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
constexpr int LIBM_TAB_PAD = (LIBM_TAB_BYTES + 15) & ~15;

// LDS ordering inside ONE wavefront: its LDS instructions execute in order, so a write by one lane is
// visible to a later read by another lane of the same wavefront; only the compiler must not reorder.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(BH_WAVE * GROUP_WPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void swd_group_kernel(SwdMultiArgs A, int Gflags, int wave_lds)
{
    const int cls = (A.split != nullptr) ? (int)blockIdx.z : 1; // 0 = the deep models of a ragged batch, 1 = the rest
    const int G = A.lanes[cls];
    const SwdTarget T = A.t[blockIdx.y];
    int J = T.look > 1 ? T.look : 1; // look-ahead: trial velocities per round (per target), one lane group each
    while (J > 1 && G * J > BH_WAVE) --J;
    // Love only: further trials INSIDE a lane group.  Its recursion is scalar (every lane of the group
    // would repeat it), so lane l runs trial l mod JL instead; only the layer terms cost JL passes.
    const int JL = (T.iwave == 1 && T.inlook > 1) ? T.inlook : 1;
    const int LPM = G * J;         // lanes per model: J groups of G lanes, group r evaluates candidate r
    const int MPW = BH_WAVE / LPM; // models per wavefront (lanes >= MPW*LPM idle along as clones of lane 0)
    extern __shared__ __align__(16) unsigned char smem_all[];
    const int lane = threadIdx.x & (BH_WAVE - 1);
    const int wave = threadIdx.x / BH_WAVE;
    const int wid = blockIdx.x * GROUP_WPB + wave; // wavefront index inside this target's row of the grid
    // the workgroup's shared copy of the libm tables, then one private region per wavefront
    // this launch's range of the processing order (see SwdMultiArgs::split)
    int lo = 0, hi = A.B;
    if (A.split != nullptr) {
        const int ndeep = A.split[0];
        if (cls == 0) hi = ndeep;
        else lo = ndeep;
    }
    if (lo + (int)blockIdx.x * GROUP_WPB * MPW >= hi) return; // whole workgroup beyond the range (grid = worst case)
    const LibmTabs LT = stage_libm_tables(smem_all, threadIdx.x, BH_WAVE * GROUP_WPB);
    __syncthreads();
    if (lo + wid * MPW >= hi) return;
    unsigned char *smem = smem_all + LIBM_TAB_PAD + (size_t)wave * wave_lds;
    const bool spare = lane >= MPW * LPM;
    const int g = spare ? 0 : lane / LPM;        // model slot inside the wave
    const int rr = spare ? 0 : (lane % LPM) / G; // which candidate this lane's group evaluates
    const int li = spare ? 0 : lane % G;         // this lane's index inside its group
    const int slot = g * J + rr;                 // group index inside the wave
    const int sidx = lo + wid * MPW + g; // position in the processing order
    const bool valid = sidx < hi;
    const int ib = valid ? (A.perm ? A.perm[sidx] : sidx) : 0;
    const int Lmax = A.rows[cls]; // LDS rows per model of this class (>= every layer count it meets)
    const int K = T.K;
    const int ifunc = T.iwave; // 1 Love, 2 Rayleigh: uniform per wavefront

    // LDS carve-up of the wavefront's region (all offsets multiples of 16 B)
    double *ca = reinterpret_cast<double *>(smem);                 // [MPW*J][Lmax][CA_STRIDE]
    double *xs = ca + (size_t)MPW * J * Lmax * CA_STRIDE;          // [11][MPW]
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
        const int b = sb < hi ? (A.perm ? A.perm[sb] : sb) : 0;
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
    double *cam = ca + (size_t)slot * Lmax * CA_STRIDE; // this group's parked layers
    const bool par5 = (G >= 5) && !(Gflags & 0x100);
    const int gbase = slot * G; // first lane of this group
    // one layer count for the whole wavefront and no water layer: the recursion needs no masking
    const bool ragged = __ballot(mmax != mtop || llw != 1) != 0ull;
    const int col = li % 5;                          // the 5-vector component this lane owns

    SearchRt S;
    S.XS = MPW;
    S.init(md, mmax, valid, T.igr, K, per, xs + g, ys + g, T.vel + (size_t)ib * T.ldv, li == 0 && rr == 0 && !spare,
           T.mode, cpl + g, cpl + (size_t)K * MPW + g);

    // per-period constants of this lane's first layer (m = li) and of the half-space: they depend
    // on omega and the model only, not on the trial phase velocity -> recomputed when omega changes
    double c_omega = -1.0, c_xka = 0.0, c_xkb = 0.0, c_gammk = 0.0, h_xka = 0.0, h_xkb = 0.0, h_gammk = 0.0;
    const bool prof = (A.neval != nullptr);
    long long tA = 0, tB = 0, tS = 0, t0 = 0, t1c = 0, t2c = 0;
    const unsigned long long w_start = prof ? wall_clock64() : 0ull, c_start = prof ? clock64() : 0ull;
    unsigned int nrounds = 0;
    while (__ballot(S.active) != 0ull) {
        ++nrounds;
        // All lanes take part in the evaluation (finished models compute on stale values).
        if (prof) t0 = clock64();
        const double omg = S.omega;
        const int cb = rr * JL + (li % JL); // the trial this lane carries through the recursion
        const double cev = (J * JL == 1) ? S.ceval : S.candidate(cb);
        const double wvno = omg / cev;
        double del;
        if (ifunc == 2) {
            double omega = omg;
            if (omega < 1.0e-4) omega = 1.0e-4;
            const double wvno2 = wvno * wvno;
            if (omega != c_omega) {
                c_omega = omega;
                if (li <= mmax - 2) {
                    const double am = md.A(li), bm = md.Bv(li);
                    c_xka = omega / am;
                    c_xkb = omega / bm;
                    const double t = bm / omega;
                    c_gammk = 2.0 * t * t;
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
                    double wvnop = wvno + xka;
                    double wvnom = fabs(wvno - xka);
                    const double ra = sqrt(wvnop * wvnom);
                    wvnop = wvno + xkb;
                    wvnom = fabs(wvno - xkb);
                    const double rb = sqrt(wvnop * wvnom);
                    const double dpth = md.D(m);
                    const double rho1 = md.R(m);
                    LayerTerms v;
                    layer_products(ra * dpth, rb * dpth, ra, rb, wvno, xka, xkb, dpth, v, LT);
                    Ca19 c;
                    rayleigh_ca19(c, wvno2, gam, gammk, rho1, v);
                    park_ca25(cam + (size_t)m * CA_STRIDE, c);
                }
            }
            // half-space E vector (surfdisp96.f:800-808), redundantly in every lane
            double e[5];
            {
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
            if (prof) t1c = clock64();
            // ---- phase B: the sequential recursion, bottom-up over the parked layers -----------
            {
                double e0[5] = {e[0], e[1], e[2], e[3], e[4]};
                DivRange dr;
                dr.reset();
                if (par5) {
                    if (ragged) rayleigh_chain_group<true, true, false>(e, cam, col, gbase, mtop, mmax, llw, dr);
                    else rayleigh_chain_group<true, false, false>(e, cam, col, gbase, mtop, mmax, llw, dr);
                } else {
                    if (ragged) rayleigh_chain_group<false, true, false>(e, cam, col, gbase, mtop, mmax, llw, dr);
                    else rayleigh_chain_group<false, false, false>(e, cam, col, gbase, mtop, mmax, llw, dr);
                }
                if (!dr.ok() && S.active) { // out-of-range operand somewhere: verbatim re-run (whole groups agree)
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
                layer_products(ra * dpth, znul, ra, znul, wvno, xka, znul, dpth, v, LT);
                const double w0 = -rho1 * v.w;
                del = v.cosp * e[0] + w0 * e[1];
            }
            wave_sync();
        } else {
            const double omega = omg;
            if (omega != c_omega) {
                c_omega = omega;
                if (li <= mmax - 2) c_xkb = omega / md.Bv(li);
                const double beta1 = md.Bv(mmax - 1);
                h_xkb = omega / beta1;
                h_gammk = 1.0 / (beta1 * beta1); // e2 of the half-space (surfdisp96.f:731)
            }
            // ---- phase A (Love): cosq, y, z, xmu per layer, for each of the group's JL trials ----------
            for (int jj = 0; jj < JL; ++jj) {
                const double wv = (JL == 1) ? wvno : omg / S.candidate(rr * JL + jj);
                for (int m = li; m <= mmax - 2; m += G) {
                    if (m >= llw - 1) {
                        const double beta1 = md.Bv(m);
                        const double rho1 = md.R(m);
                        const double dm = md.D(m);
                        const double xmu = rho1 * beta1 * beta1;
                        const double xkb = (m == li) ? c_xkb : omega / beta1;
                        const double wvnop = wv + xkb;
                        const double wvnom = fabs(wv - xkb);
                        const double rb = sqrt(wvnop * wvnom);
                        const double q = dm * rb;
                        double cosq, y, z;
                        if (wv < xkb) {
                            double sinq;
                            bh_sincos(q, &sinq, &cosq, LT);
                            y = sinq / rb;
                            z = -rb * sinq;
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
                        dst[2] = make_double2(bh_rcp_refined(xmu), 0.0);
                    }
                }
            }
            double e1, e2;
            {
                const double rho1 = md.R(mmax - 1);
                const double xkb = h_xkb;
                const double wvnop = wvno + xkb;
                const double wvnom = fabs(wvno - xkb);
                const double rb = sqrt(wvnop * wvnom);
                e1 = rho1 * rb;
                e2 = h_gammk;
            }
            wave_sync();
            if (prof) t1c = clock64();
            {
                const double s1 = e1, s2 = e2;
                DivRange dr;
                dr.reset();
                const double *camt = cam + LOVE_TERMS * (li % JL); // this lane's trial
                if (ragged) love_chain_group<true, false>(e1, e2, camt, mtop, mmax, llw, dr);
                else love_chain_group<false, false>(e1, e2, camt, mtop, mmax, llw, dr);
                if (!dr.ok() && S.active) {
                    e1 = s1;
                    e2 = s2;
                    love_chain_group<true, true>(e1, e2, camt, mtop, mmax, llw, dr);
                }
            }
            del = e1;
            wave_sync();
        }
        if (prof) t2c = clock64();
        // Every lane of the model can read all J (velocity, value) pairs; the search consumes them for
        // as long as its next request is the very velocity (at the same omega) the next group evaluated.
        {
            bool live = S.active;
            const int Jtot = J * JL;
            for (int j = 0; j < Jtot; ++j) {
                double dj = del;
                if (Jtot > 1) {
                    const int src = (g * J + j / JL) * G + (j % JL); // a lane that carried trial j
                    const double cj = __shfl(cev, src);
                    dj = __shfl(del, src);
                    // trial 0 IS the pending request (consumed unconditionally, also when a broken model
                    // has driven the search to NaN); a later trial only if the search now asks for it
                    if (j > 0) live = live && S.active && S.ceval == cj && S.omega == omg;
                }
                if (__ballot(live) == 0ull) break;
                if (live) S.advance(dj);
            }
        }
        if (prof) {
            const long long t3 = clock64();
            tA += t1c - t0;
            tB += t2c - t1c;
            tS += t3 - t2c;
        }
    }
    if (valid && li == 0 && rr == 0 && !spare) T.err[ib] = S.errflag;
    if (prof) {
        unsigned long long tot = (li == 0 && rr == 0 && !spare) ? S.evals : 0u;
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        if (lane == 0) {
            atomicAdd(A.neval, tot);
            // development aid: wave-cycles per phase, [1..3] Rayleigh A/B/state, [4..6] Love
            const int o = (ifunc == 2) ? 1 : 4;
            atomicAdd(A.neval + o, (unsigned long long)tA);
            atomicAdd(A.neval + o + 1, (unsigned long long)tB);
            atomicAdd(A.neval + o + 2, (unsigned long long)tS);
            const unsigned long long widx = atomicAdd(A.neval + 7, 1ull);
            if (widx < BH_TRACE_WAVES) { // development aid: one record per wavefront (tools/gpu_trace.py)
                unsigned long long *r = A.neval + BH_COUNTER_WORDS + 4 * widx;
                unsigned hwid, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                hwid = (hwid & 0xffffu) | ((xcc & 0xfu) << 16);
                r[0] = w_start;
                r[1] = wall_clock64();
                r[2] = clock64() - c_start;
                r[3] = (unsigned long long)nrounds | ((unsigned long long)ifunc << 32) | ((unsigned long long)hwid << 36);
            }
        }
    }
}

#include "swd_group2.inc"

size_t group_lds_bytes(int G, int J, int Lmax, int Kmax, int maxmode)
{
    const int MPW = BH_WAVE / (G * J);
    return ((size_t)MPW * J * Lmax * CA_STRIDE + (size_t)2 * NEV_MAX * MPW +
            (size_t)((Kmax + 1) & ~1)) * sizeof(double) +
           (((size_t)4 * Lmax * MPW * sizeof(float) + 15) & ~(size_t)15) +
           (maxmode > 1 ? (size_t)2 * Kmax * MPW * sizeof(double) : 0);
}

// ---- processing order: models by layer count, deepest first (counting sort, one workgroup) -------------
__global__ __launch_bounds__(1024) void order_kernel(int B, const int32_t *nlay, int32_t *perm, int Lcut, int32_t *split)
{
    __shared__ int bin[BH_MAX_LAYERS + 2];
    const int tid = threadIdx.x;
    for (int i = tid; i < BH_MAX_LAYERS + 2; i += 1024) bin[i] = 0;
    __syncthreads();
    for (int b = tid; b < B; b += 1024) {
        int n = nlay[b];
        n = n < 0 ? 0 : (n > BH_MAX_LAYERS + 1 ? BH_MAX_LAYERS + 1 : n);
        atomicAdd(&bin[n], 1);
    }
    __syncthreads();
    if (tid == 0) { // start offset of every depth, deepest first
        int acc = 0;
        for (int n = BH_MAX_LAYERS + 1; n >= 0; --n) {
            const int c = bin[n];
            bin[n] = acc;
            acc += c;
            if (n == Lcut + 1 && split != nullptr) split[0] = acc; // models with more than Lcut layers come first
        }
        if (split != nullptr && Lcut + 1 > BH_MAX_LAYERS + 1) split[0] = 0;
    }
    __syncthreads();
    for (int b = tid; b < B; b += 1024) {
        int n = nlay[b];
        n = n < 0 ? 0 : (n > BH_MAX_LAYERS + 1 ? BH_MAX_LAYERS + 1 : n);
        perm[atomicAdd(&bin[n], 1)] = b; // order inside a depth is arbitrary: per-model results do not depend on it
    }
}

// log / powf of the host's libm (what the reference's compiled Fortran calls), restated in bh_libm.h;
// arguments outside the restated paths (never produced by a physical model) use the device library.
__device__ __forceinline__ double sphere_log(double x)
{
    double y;
    return bhp_log(x, &y, bhp_log_data) ? y : log(x);
}
__device__ __forceinline__ float sphere_powf(float x, float y)
{
    float r;
    return bhp_powf(x, y, &r, bhp_powf_log2_data, bhp_exp2f_data) ? r : powf(x, y);
}

// ---- earth flattening, surfdisp96.f:486-553 (`sphere`, both calls) --------------------------------
// One lane per model.  Arithmetic widths as in the Fortran (model arrays binary32, radii binary64);
// log() and powf() are the glibc restatements, so the transformed model is bit-identical to the reference's.
__global__ void sphere_kernel(int B, int Lmax, const int32_t *nlay, const double *h, const double *vp,
                              const double *vs, const double *rho, ptrdiff_t sl, ptrdiff_t sb, double *oh,
                              double *ovp, double *ovs, double *orl, double *orr)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int mmax = nlay[b];
    const double ar = 6370.0;
    double dr = 0.0, r0 = ar;
    for (int i = 0; i < mmax; ++i) {
        const ptrdiff_t o = (ptrdiff_t)b * sb + (ptrdiff_t)i * sl;
        const float d = (i == mmax - 1) ? 1.0f : (float)h[o]; // d(mmax) = 1.0 while transforming
        const float a = (float)vp[o], bb = (float)vs[o], rt = (float)rho[o];
        dr = dr + (double)d;
        const double r1 = ar - dr;
        const double z0 = ar * sphere_log(ar / r0);
        const double z1 = ar * sphere_log(ar / r1);
        const float dn = (i == mmax - 1) ? 0.0f : (float)(z1 - z0); // d(mmax) = 0 afterwards
        const double tmp = (ar + ar) / (r0 + r1);                   // layer mid-point
        const float btp = (float)tmp;
        float p5 = btp * btp; // btp**(-5) the way compiler-rt's __powisf2 does it
        p5 = p5 * p5;
        p5 = btp * p5;
        const size_t q = (size_t)i * B + b;
        oh[q] = (double)dn;
        ovp[q] = (double)(float)((double)a * tmp);
        ovs[q] = (double)(float)((double)bb * tmp);
        orl[q] = (double)(rt * (1.0f / p5));
        orr[q] = (double)(rt * sphere_powf(btp, -2.275f));
        r0 = r1;
    }
}

// ---- np.interp, row-wise (numpy/core/src/multiarray/compiled_base.c: arr_interp) -------------------
__global__ void interp_kernel(int B, int K0, const double *x0, const double *y0, int ld0, int K1,
                              const double *x1, double *y1, int ld1)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * K1) return;
    const int b = t / K1, k = t % K1;
    const double x = x1[k];
    const double *fp = y0 + (size_t)b * ld0;
    double r;
    if (x < x0[0]) r = fp[0];
    else if (x > x0[K0 - 1]) r = fp[K0 - 1];
    else {
        int j = 0; // largest j with x0[j] <= x
        for (int lo = 0, hi = K0; lo < hi;) {
            const int mid = (lo + hi) >> 1;
            if (x0[mid] <= x) { j = mid; lo = mid + 1; } else hi = mid;
        }
        if (j == K0 - 1 || x0[j] == x) r = fp[j];
        else {
            const double slope = (fp[j + 1] - fp[j]) / (x0[j + 1] - x0[j]);
            r = slope * (x - x0[j]) + fp[j];
            if (r != r) { // numpy's nan rescue (an infinity in fp)
                r = slope * (x - x0[j + 1]) + fp[j + 1];
                if (r != r && fp[j] == fp[j + 1]) r = fp[j];
            }
        }
    }
    y1[(size_t)b * ld1 + k] = r;
}

} // namespace

void bh_launch_order(int B, const int32_t *nlay, int32_t *perm, int Lcut, int32_t *split, hipStream_t stream)
{
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, stream, B, nlay, perm, Lcut, split);
}

void bh_launch_sphere(int B, int Lmax, const int32_t *nlay, const double *h, const double *vp,
                      const double *vs, const double *rho, ptrdiff_t sl, ptrdiff_t sb, double *oh,
                      double *ovp, double *ovs, double *orho_love, double *orho_ray, hipStream_t stream)
{
    hipLaunchKernelGGL(sphere_kernel, dim3((B + 127) / 128), dim3(128), 0, stream, B, Lmax, nlay, h, vp, vs, rho,
                       sl, sb, oh, ovp, ovs, orho_love, orho_ray);
}

void bh_launch_interp(int B, int K0, const double *x0, const double *y0, int ld0, int K1,
                      const double *x1, double *y1, int ld1, hipStream_t stream)
{
    const int n = B * K1;
    hipLaunchKernelGGL(interp_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, B, K0, x0, y0, ld0, K1, x1, y1, ld1);
}

size_t bh_swd_lds_bytes(int Lmax, int K, int mode)
{
    return (size_t)4 * Lmax * BH_WAVE * sizeof(float) + (size_t)2 * NEV_MAX * BH_WAVE * sizeof(double) +
           (size_t)((K + 1) & ~1) * sizeof(double) + LIBM_TAB_BYTES +
           (mode > 1 ? (size_t)2 * K * BH_WAVE * sizeof(double) : 0);
}

void bh_launch_swd(const SwdKernelArgs &a, int iwave, hipStream_t stream)
{
    const int grid = (a.B + BH_WAVE - 1) / BH_WAVE;
    const size_t lds = bh_swd_lds_bytes(a.Lmax, a.K, a.mode);
    if (iwave == 1)
        hipLaunchKernelGGL(swd_kernel<1>, dim3(grid), dim3(BH_WAVE), lds, stream, a);
    else
        hipLaunchKernelGGL(swd_kernel<2>, dim3(grid), dim3(BH_WAVE), lds, stream, a);
}

// ---- launch plan ------------------------------------------------------------------------------------
// Which mapping (one lane per model, or G lanes per model) and how many trial velocities per round and
// target.  Cost model, calibrated on MI355X with 10-layer models and 30 periods (profiles/):
// at this kernel's register budget 2 wavefronts are resident per SIMD = 2048 on the chip; a launch runs in
// ceil(wavefronts / 2048) rounds, each as long as its longest wavefront.  Relative wavefront durations:
//   group kernel, Rayleigh: 1 / .64 / .52 / .45 / .36 for 1 / 2 / 3 / 4 / 7 trials per round,
//   group kernel, Love:     .735 / .54 / .39 / .33 / .26 (with two in-group trials at the first two levels),
//   one lane per model:     4.3 (64 models per wavefront, every layer term serial).
// More trials shorten every model's chain of dependent secular evaluations (what a small batch is bound
// by) but cost lanes, i.e. wavefronts.  The plan minimises rounds x longest wavefront.
namespace {
constexpr int PLAN_LEVELS = 5;
const int plan_trials[PLAN_LEVELS] = {1, 2, 3, 4, 7};
const double plan_dur[2][PLAN_LEVELS] = {{0.735, 0.54, 0.39, 0.33, 0.26}, {1.0, 0.64, 0.52, 0.45, 0.36}};
constexpr int PLAN_SLOTS = 2048;
constexpr double PLAN_LANE_PER_MODEL = 4.3;

int plan_fit(int G, int level) // largest trial count <= the level's that fits a wavefront
{
    int J = plan_trials[level];
    while (J > 1 && G * J > BH_WAVE) --J;
    return J;
}
} // namespace

int bh_swd_pick_group(int B, int ntargets, int Lmax)
{
    // one lane per finite layer (at least 5: the width of the Rayleigh vector recursion), capped at
    // 16, so that phase A needs a single round
    int G = Lmax - 1;
    if (G < 5) G = 5;
    if (G > 16) G = 16;
    (void)B; (void)ntargets;
    return G;
}

// Returns the plan's cost; *G = 1 selects the lane-per-model kernel (look[] is then all 1).
// Gforce > 0: the caller fixed the lanes per model, only the trials are planned.
double bh_swd_plan(int B, int Lmax, int ntargets, const int *iwave, int Gforce, int *G, int *look)
{
    const int Gg = Gforce > 1 ? Gforce : bh_swd_pick_group(B, ntargets, Lmax);
    int lvl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, best_lvl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double best = 1e300;
    auto cost = [&](const int *l) {
        long waves = 0;
        double dmax = 0.0, dsum = 0.0;
        for (int t = 0; t < ntargets; ++t) {
            const int J = plan_fit(Gg, l[t]);
            const int mpw = BH_WAVE / (Gg * J);
            waves += (B + mpw - 1) / mpw;
            const double d = plan_dur[iwave[t] == 2 ? 1 : 0][l[t]];
            dmax = d > dmax ? d : dmax;
            dsum += d;
        }
        const long rounds = (waves + PLAN_SLOTS - 1) / PLAN_SLOTS;
        // (two wavefronts on a SIMD slow each other down a little; ties go to the shorter wavefronts)
        return (double)rounds * dmax * (waves > PLAN_SLOTS / 2 ? 1.15 : 1.0) + 1e-3 * dsum;
    };
    if (ntargets <= 4) { // exhaustive
        for (;;) {
            const double c = cost(lvl);
            if (c < best) {
                best = c;
                for (int t = 0; t < ntargets; ++t) best_lvl[t] = lvl[t];
            }
            int t = 0;
            while (t < ntargets && ++lvl[t] == PLAN_LEVELS) lvl[t++] = 0;
            if (t == ntargets) break;
        }
    } else { // same level for all targets
        for (int l = 0; l < PLAN_LEVELS; ++l) {
            for (int t = 0; t < ntargets; ++t) lvl[t] = l;
            const double c = cost(lvl);
            if (c < best) {
                best = c;
                for (int t = 0; t < ntargets; ++t) best_lvl[t] = l;
            }
        }
    }
    // the lane-per-model kernel: one launch per target, 64 models per wavefront
    const long w1 = (long)ntargets * ((B + BH_WAVE - 1) / BH_WAVE);
    const double c1 = (double)((w1 + PLAN_SLOTS - 1) / PLAN_SLOTS) * PLAN_LANE_PER_MODEL;
    if (Gforce == 1 || (Gforce <= 0 && c1 < best)) {
        *G = 1;
        for (int t = 0; t < ntargets; ++t) look[t] = 1;
        return c1;
    }
    *G = Gg;
    for (int t = 0; t < ntargets; ++t) look[t] = plan_fit(Gg, best_lvl[t]);
    return best;
}

// LDS of one workgroup = shared libm tables + GROUP_WPB wavefront regions
size_t bh_swd_group_lds_bytes(int G, int J, int Lmax, int Kmax, int maxmode)
{
    return LIBM_TAB_PAD + GROUP_WPB * ((group_lds_bytes(G, J, Lmax, Kmax, maxmode) + 15) & ~(size_t)15);
}

// A wavefront's LDS region small enough for 8 wavefronts (4 workgroups) per CU of 160 KB
constexpr size_t WAVE_LDS_TARGET = (160 * 1024 / 4 - LIBM_TAB_PAD) / GROUP_WPB;
constexpr size_t WG_LDS_CAP = 64 * 1024;

int bh_launch_swd_group(const SwdMultiArgs &a0, int G0, hipStream_t stream)
{
    int kmax = 0, maxmode = 1;
    for (int t = 0; t < a0.ntargets; ++t) {
        kmax = a0.t[t].K > kmax ? a0.t[t].K : kmax;
        maxmode = a0.t[t].mode > maxmode ? a0.t[t].mode : maxmode;
    }
    static const int redundant = std::getenv("BH_SWD_REDUNDANT") ? 0x100 : 0; // experiment switch
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
            bool same = waves(Gwide) <= PLAN_SLOTS / 8;
            for (int t = 0; t < a.ntargets; ++t) same = same && trials(Gwide, t) == trials(G, t);
            if (same) {
                G = Gwide;
                break;
            }
        }
        while (most_models(G) > 1 && wave_bytes(G) > WAVE_LDS_TARGET) G += 1;
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
    const dim3 grid((nwaves + GROUP_WPB - 1) / GROUP_WPB, a.ntargets, two ? 2 : 1);
    const size_t lds = LIBM_TAB_PAD + GROUP_WPB * wave_lds;
    hipLaunchKernelGGL(swd_group_kernel, grid, dim3(BH_WAVE * GROUP_WPB), lds, stream, a, redundant, (int)wave_lds);
    return 0;
}

// ---- group kernel v2 (swd_group2.inc): Love and Rayleigh targets are separate launches (own register budget,
// own LDS size), the Love launch goes to `stream_love` between ev_fork / ev_join so that both run side by side.
size_t bh_swd_group2_lds_bytes(int G, int J, int rows, int Kmax, int maxmode, int iwave)
{
    return LIBM_TAB_PAD + G2_PER_BYTES + g2_shared_lds_bytes(G, J, Kmax, maxmode, iwave) + G2_WPB * g2_wave_lds_bytes(G, J, rows, iwave);
}

namespace {
// per wavefront, so that three Rayleigh workgroups (or two + two of Love's smaller ones) share a CU's 160 KB
constexpr size_t G2_WAVE_LDS_TARGET = 160 * 1024 / 3 / G2_WPB; // a wavefront's share of its workgroup's LDS
constexpr size_t G2_WG_LDS_CAP = 160 * 1024;

int launch_group2_family(const SwdMultiArgs &a0, int iwave, int G0, hipStream_t stream)
{
    SwdMultiArgs a = a0;
    a.ntargets = 0;
    int kmax = 0, maxmode = 1;
    for (int t = 0; t < a0.ntargets; ++t)
        if (a0.t[t].iwave == iwave) {
            a.t[a.ntargets++] = a0.t[t];
            kmax = a0.t[t].K > kmax ? a0.t[t].K : kmax;
            maxmode = a0.t[t].mode > maxmode ? a0.t[t].mode : maxmode;
        }
    if (a.ntargets == 0) return 0;
    const bool two = a0.split != nullptr && a0.Lcut < a0.Lmax;
    if (!two) a.split = nullptr;
    size_t wave_lds = 0, shared_lds = 0;
    int nwaves = 1;
    static const size_t lds_target = std::getenv("BH_SWD_LDS_TARGET") ? (size_t)std::atol(std::getenv("BH_SWD_LDS_TARGET")) : G2_WAVE_LDS_TARGET; // experiment switch
    for (int cls = two ? 0 : 1; cls <= 1; ++cls) {
        const int rows = (two && cls == 1) ? a0.Lcut : a0.Lmax;
        int G = G0;
        auto trials = [&](int g, int t) {
            int J = a.t[t].look > 1 ? a.t[t].look : 1;
            while (J > 1 && g * J > BH_WAVE) --J;
            return J;
        };
        auto wave_bytes = [&](int g) { // a wavefront's own block
            size_t w = 0;
            for (int t = 0; t < a.ntargets; ++t) {
                const size_t l = g2_wave_lds_bytes(g, trials(g, t), rows, iwave);
                w = l > w ? l : w;
            }
            return w;
        };
        auto shared_bytes = [&](int g) { // the workgroup's shared block
            size_t w = 0;
            for (int t = 0; t < a.ntargets; ++t) {
                const size_t l = g2_shared_lds_bytes(g, trials(g, t), kmax, maxmode, iwave);
                w = l > w ? l : w;
            }
            return w;
        };
        auto wg_bytes = [&](int g) { return LIBM_TAB_PAD + G2_PER_BYTES + shared_bytes(g) + G2_WPB * wave_bytes(g); };
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
        // a batch that leaves most of the chip idle anyway: one lane per layer of the deepest model of the class
        for (int Gwide = rows - 1 > 16 ? 16 : rows - 1; Gwide > G; --Gwide) {
            bool same = waves(Gwide) <= 512;
            for (int t = 0; t < a.ntargets; ++t) same = same && trials(Gwide, t) == trials(G, t);
            if (same) {
                G = Gwide;
                break;
            }
        }
        // fewer models per wavefront (more lanes per model) until the parked layers fit: first the residency
        // target, at the latest a CU's 160 KB
        while (most_models(G) > 1 && wg_bytes(G) > G2_WPB * lds_target) G += 1;
        while (G < BH_WAVE && wg_bytes(G) > G2_WG_LDS_CAP) G += 1;
        if (wg_bytes(G) > G2_WG_LDS_CAP) return -1;
        a.rows[cls] = rows;
        a.lanes[cls] = G;
        const size_t wb = wave_bytes(G), sb = shared_bytes(G);
        wave_lds = wb > wave_lds ? wb : wave_lds;
        shared_lds = sb > shared_lds ? sb : shared_lds;
        for (int t = 0; t < a.ntargets; ++t) {
            const int mpw = BH_WAVE / (G * trials(G, t));
            const int nx = (a.B + mpw - 1) / mpw;
            nwaves = nx > nwaves ? nx : nwaves;
        }
    }
    if (!two) {
        a.rows[0] = a.rows[1];
        a.lanes[0] = a.lanes[1];
    }
    const dim3 grid((nwaves + G2_WPB - 1) / G2_WPB, a.ntargets, two ? 2 : 1);
    const size_t lds = LIBM_TAB_PAD + G2_PER_BYTES + shared_lds + G2_WPB * wave_lds;
    if (lds > 64 * 1024) { // beyond the default limit of dynamic LDS per workgroup: opt in (deep models)
        static size_t allowed[3] = {0, 0, 0};
        if (lds > allowed[iwave]) {
            const hipError_t he = (iwave == 1)
                ? hipFuncSetAttribute(reinterpret_cast<const void *>(swd_group2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G2_WG_LDS_CAP)
                : hipFuncSetAttribute(reinterpret_cast<const void *>(swd_group2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G2_WG_LDS_CAP);
            if (he != hipSuccess) return -1;
            allowed[iwave] = G2_WG_LDS_CAP;
        }
    }
    if (iwave == 1)
        hipLaunchKernelGGL(swd_group2_kernel<1>, grid, dim3(BH_WAVE * G2_WPB), lds, stream, a, (int)shared_lds, (int)wave_lds);
    else
        hipLaunchKernelGGL(swd_group2_kernel<2>, grid, dim3(BH_WAVE * G2_WPB), lds, stream, a, (int)shared_lds, (int)wave_lds);
    return 0;
}
} // namespace

int bh_launch_swd_group2(const SwdMultiArgs &a, int G, hipStream_t stream, hipStream_t stream_love,
                         hipEvent_t ev_fork, hipEvent_t ev_join)
{
    bool love = false, ray = false;
    for (int t = 0; t < a.ntargets; ++t) {
        love = love || a.t[t].iwave == 1;
        ray = ray || a.t[t].iwave == 2;
    }
    const bool fork = love && ray && stream_love != nullptr && stream_love != stream;
    if (fork) {
        if (hipEventRecord(ev_fork, stream) != hipSuccess || hipStreamWaitEvent(stream_love, ev_fork, 0) != hipSuccess) return -2;
    }
    if (launch_group2_family(a, 2, G, stream) != 0) return -1;
    if (launch_group2_family(a, 1, G, fork ? stream_love : stream) != 0) return -1;
    if (fork) {
        if (hipEventRecord(ev_join, stream_love) != hipSuccess || hipStreamWaitEvent(stream, ev_join, 0) != hipSuccess) return -2;
    }
    return 0;
}
