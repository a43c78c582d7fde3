// bayhunter_amd/csrc/swd_common.h -- device code shared by the dispersion kernels (swd_kernel.hip: one lane per
// model; swd_group_kernel.hip: G lanes per model): the glibc-exact libm front-end, the layer terms and recursions
// of the Love / Rayleigh secular functions (surfdisp96.f:710-1068) and the per-model search state machine
// (surfdisp96.f:172-357, :390-482, :557-686).  Included inside each file's anonymous namespace.
#pragma once

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
    // Magnitudes are tracked through the HIGH dword of the (non-negative) operand: for non-negative doubles
    // that word orders like the value, and two 32-bit integer min/max per operand cost a quarter of the two f64
    // ones (which also drag a canonicalisation of the loop-carried bound along).  The bounds are one binade
    // inside [2^-400, 2^400]; zero, denormals, infinities and NaNs (as operand or as bound) fail the test.
    unsigned lo, hi;
    __device__ __forceinline__ void reset() { lo = 0x3ff00000u; hi = 0x3ff00000u; }
    __device__ __forceinline__ void see(double a) // a >= 0
    {
        const unsigned w = (unsigned)__double2hiint(a);
        lo = min(lo, w);
        hi = max(hi, w);
    }
    __device__ __forceinline__ bool ok() const
    {
        return lo >= 0x27000000u /* 2^-399 */ && hi < 0x58f00000u /* 2^400 */;
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

// ---- the Love MODE COUNT (not in the reference; oracle/swd_oracle.c: bho_dltar1_count restates it) -----------------
// The SH problem at fixed omega is a Sturm-Liouville problem in k^2.  With (e1, e2) ~ (stress, displacement) of the solution
// that decays in the half-space (c < beta of the half-space), integrated upward as dltar1 does,
//     N(c) = number of sign changes of dltar1(., omega) below c = Z + [e1 * e2 < 0 at the surface],
// Z = zeros of the displacement inside the finite layers (oscillation theorem: they enter at the free surface one by one,
// a Dirichlet eigenvalue between two Neumann ones).  Inside ONE layer the displacement is a pure sinusoid of phase advance
// q = d * rb (or a cosh / sinh combination: at most one zero), so the layer holds floor(q / pi) zeros or one more, and which
// of the two is the sign change of e2 across the layer -- taken from the recursion's own values.  So sign(dltar1) ==
// (-1)^N exactly as computed, and N(c2) - N(c1) says how many sign changes the reference's function has between two trial
// velocities WITHOUT visiting the grid points in between (SearchT: the counted scan).
// love_zero_floor: floor(q / pi) of a propagating layer as a double, -1 where it is ambiguous by two (q / pi within 1e-9 of
// an integer: a zero at both ends of the layer at once); 0 for an evanescent layer.
__device__ __forceinline__ double love_zero_floor(double q)
{
    const double x = q * 0.31830988618379067154, xf = floor(x);
    const bool ok = (x - xf > 1.0e-9) && (xf + 1.0 - x > 1.0e-9) && (x < 1.0e9);
    return ok ? xf : -1.0;
}
struct LoveCount {
    int n;
    bool ok;
    __device__ __forceinline__ void reset(bool half_space_decays)
    {
        n = 0;
        ok = half_space_decays;
    }
    // one layer: fl from love_zero_floor, the displacement before (e2_old) and after (e2_new) the layer
    __device__ __forceinline__ void layer(double fl, double e2_old, double e2_new)
    {
        const int f = (int)fl;
        ok = ok && !(fl < 0.0);
        n += f + ((((f & 1) != 0) != signs_differ(e2_new, e2_old)) ? 1 : 0);
    }
    __device__ __forceinline__ int total(double e1, double e2) const { return n + (signs_differ(e1, e2) ? 1 : 0); }
    // packed for the exchange between lanes: count, or -1 when not valid
    __device__ __forceinline__ int packed(double e1, double e2) const { return ok ? total(e1, e2) : -1; }
};

// ---- Love: SH Thomson-Haskell (surfdisp96.f:710-769) ----------------------------------------
// nv (optional): the packed mode count (LoveCount::packed) of this evaluation
template <bool EXACT>
__device__ double love_secular(double wvno, double omega, const ModelLds &md, int mmax, int llw,
                               int mtop, DivRange &dr, const LibmTabs &LT, int *nv = nullptr)
{
    double beta1 = md.Bv(mmax - 1);
    double rho1 = md.R(mmax - 1);
    double xkb = omega / beta1;
    double wvnop = wvno + xkb;
    double wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double e1 = rho1 * rb;
    double e2 = 1.0 / (beta1 * beta1);
    LoveCount lc;
    lc.reset(wvno > xkb);
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
            double cosq, y, z, fl = 0.0;
            if (wvno < xkb) {
                double sinq;
                bh_sincos(q, &sinq, &cosq, LT);
                y = sinq / rb;
                z = -rb * sinq;
                fl = love_zero_floor(q);
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
            const double e2o = e2;
            love_step<EXACT>(e1, e2, cosq, y, z, xmu, EXACT ? 0.0 : bh_rcp_refined(xmu), dr);
            lc.layer(fl, e2o, e2);
        }
    }
    if (nv != nullptr) *nv = lc.packed(e1, e2);
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

#include "swd_fa.h"

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
    ST_NEVF = 4,  // forced midpoint after the estimate left the bracket (:594-598)
    // the short refinement (SearchT<.., FAST = true>; not the reference's sequence, see there)
    ST_FX = 5,    // single point: regula falsi / bisection
    ST_FP1 = 6,   // x - tau (towards c1) of the acceptance pair around the estimate x
    ST_FP2 = 7,   // x + tau (towards c2)
    // (8: unused)
    // the counted scan (Love; see SearchT): not the reference's sequence of scan evaluations, but the same bracket
    ST_JUMP = 9,  // the grid point `jn` steps above c1
    ST_BIS = 10,  // a grid point between c1 and c2 (`jn` steps above c1; c2 is `nn` steps above c1 and counts more)
    // probes of the guard of the short refinement (FAST only; see SearchT)
    ST_GH = 11,   // just above the accepted bracket's upper end
    ST_GL = 12,   // just below its lower end
    ST_GS1 = 13,  // just inside the lower end of a scan step that contains a half-space velocity and showed no sign change
    ST_GS2 = 14   // just inside its upper end
};

// ---- the per-model search state machine -------------------------------------------------------
// Everything the reference's driver (surfdisp96.f:172-357), getsol (:390-482) and nevill
// (:557-686) keep between two secular-function evaluations, for the fundamental mode.
// `advance(del)` consumes the value of the secular function at `ceval` and either finishes the
// model or leaves the next phase velocity to evaluate in `ceval` (with `omega` current).
// NLO: Neville entries kept in LDS; the orders from NLO on (rarely reached: the order only grows through
// consecutive interpolation steps) live in a second array (`set_high`, global memory in the lane-per-evaluation
// kernel, whose residency is bounded by LDS).  NLO = NEV_MAX: everything in LDS, no second array.
// FAST = true compiles in the engine's short refinement (bh_engine_set_swd_search; the engine's default since ABI 7), taken by phase-velocity
// targets: the bracket scan in steps of dc is the reference's, evaluation for evaluation -- the same bracket, the same
// root -- but inside the bracket nevill's 10-12 evaluations (its stop test is the bracket WIDTH, which one-sided
// interpolation steps close slowly) are replaced by ~3: one regula-falsi point, then an inverse-quadratic estimate x
// through the three known points, accepted as soon as the function changes sign between x - tau and x + tau
// (tau = 5e-8 |x|; the reference stops at a bracket of 1e-6 c1 and returns one of its ends).  A miss moves the
// bracket and repeats; bisection from the seventh pass on.  Result: within 1.2e-6 relative of the reference's
// (measured: tests/test_gpu_swd_fast.py), against north_star's 1e-5; NOT bit-identical (BH_SEARCH_REFERENCE is).  The sequence
// of evaluations is a function of the model alone (not of the launch plan): results do not depend on the batch.
// oracle/swd_oracle.c (refine_root_fast) restates it for the bit-level check of the device.
// FASTM: 0 = the reference sequence only; 1 = FAST as described (group-velocity targets fall through to the reference
// sequence); 2 = FAST for launches WITHOUT group-velocity targets: nevill and the second-root logic are not compiled in.
// SIMPLE: the launch holds fundamental-mode phase-velocity targets only (the usual inversion set-up): the second root of a
// group velocity and the mode loop are not compiled in (fewer live registers in the round loop).
//
// THE COUNTED SCAN (Love targets, every build; bh_engine_set_swd_scan, on by default).  getsol looks for the first step
// [g, g + dc] of its grid g_i = g_(i-1) + dc over which the secular function changes sign.  With the Love mode count N
// (LoveCount above: sign f == (-1)^N as computed) the grid points need not all be visited: N(g_(i+s)) == N(g_i) proves that
// none of the s steps in between shows a sign change -- the reference would walk through them --, a larger count proves there is
// one, and a search over the grid INDEX finds the lowest step whose upper end counts more than g_i (taken as the bracket when
// the two ends differ in sign, walked over when they do not: two roots inside one step, invisible to the reference as
// well).  Every grid point is formed by the reference's own repeated additions of dc, so the bracket handed to the
// refinement -- and every bit after it -- is the reference's: same velocities, same failure flags, a third of the scan's
// evaluations.  Upward scans below min(half-space S velocity, betmx) only; everything else (downward scans, a scan that
// turned round at clow, an ambiguous count, Rayleigh: no such count for the P-SV problem here) takes the reference's steps.
// The stride is a function of the search state alone: the first aims one step beyond where the last period's bracket was
// found, later ones 4, 8, ... 64 (first period: 16, 32, 64); the index search is a regula falsi when exactly one sign
// change lies between the ends, a bisection otherwise.  oracle/swd_oracle.c (bracket_and_refine, scan mode 1) restates it.
//
// THE GUARD of the short refinement (FAST, phase velocities).  The reference's scan grid is anchored at the PREVIOUS root
// (c1 = c(k-1) - 1.5 dc), which the short sequence knows to 1.05e-6 relative only: the two grids differ by s, and the two
// scans see different sign patterns exactly when a sign change lies within |s| of a grid point.  With a lone root that moves
// the bracket by a step (same root); but the half-space terms contain |k - k_v|, so a root creeping up to a half-space
// velocity v has a mirror-image sign change just above v, and with that pair it decides whether the reference sees a sign
// change AT ALL (docs/HISTORY.md 3.1b: the only mechanism in 9.4 million random models).  The guard, eps = 3e-6 x velocity:
//   - a scan step without a sign change that contains a half-space velocity may hide the pair: probe eps inside both ends;
//   - an accepted bracket whose root lies within two steps of a half-space velocity / betmx: root within eps of a bracket
//     end or of betmx, or a sign change within eps OUTSIDE a bracket end (the image right behind it).
// A model the guard fires on (`guard`) is run again with the REFERENCE's sequence by the engine (a second, small launch):
// failure flags and zero-from-period-k rows are then the reference's by construction.
template <int XSC, int NLO = NEV_MAX, int FASTM = 0, bool SIMPLE = false> // XSC > 0: compile-time lane stride of the Neville tables in LDS; 0: run-time (member XS)
struct SearchT {
    static constexpr bool FAST = FASTM != 0, PHASE_ONLY = FASTM == 2, NOGROUP = PHASE_ONLY || SIMPLE;
    // counted scan
    static constexpr int stride_first = 16, stride_next = 4, stride_max = 64;
    // (per-lane flags share ONE register: a `bool` member lives as a 64-bit lane mask in a scalar register pair, and this
    //  kernel has none to spare -- six more of them cost the round loop 30 spilled SGPRs)
    enum : unsigned { F_CNT_ON = 1u, F_CNT_OK = 2u, F_JUMP_READY = 4u, F_GUARD_ON = 8u, F_GUARD = 16u, F_FLO_NEG = 32u,
                      F_SEED3 = 64u,   // Rayleigh, short refinement: the scan's last replaced point seeds the first estimate
                      F_SEEDED = 128u, // this bracket's refinement started with a third point
                      F_FA = 512u,     // the values come from the fast arithmetic (swd_fa.h): values that are not numbers (below fa::SIGN_FLOOR) fire the guard
                      // A group velocity's two chains of roots (igr = 2 / enter_second below): the root at t/(1+h) of period k + 1 starts
                      // from the root at t/(1+h) of period k (:262-266), the root at t/(1-h) from the root at t/(1+h) of its own period
                      // (:282-287) -- and nothing starts from it.  The first roots are a chain of K dependent searches, the second
                      // roots K independent ones: a launch of the first chain (F_CHAIN_A: the periods t/(1+h), every root stored
                      // unrounded in the output row), then a launch of one search per (model, period) (F_ONE) that reads the first
                      // root there and puts the group velocity in its place.  Same searches, same values, same bits as one after the
                      // other; the chain of dependent roots is half as long (not compiled into the builds without group velocities).
                      F_CHAIN_A = 1024u, F_ONE = 2048u
    };
    // (F_GUARD_ON doubles as "this search takes the short refinement": in a build with both sequences (FASTM = 1) a phase-velocity
    //  target can be told to keep the reference's -- init(.., refseq) --, e.g. the Love targets under BH_SEARCH_FAST_RAYLEIGH)
    unsigned flg = 0u;
    __device__ __forceinline__ bool has(unsigned f) const { return (flg & f) != 0u; }
    __device__ __forceinline__ void put(unsigned f, bool v) { flg = v ? (flg | f) : (flg & ~f); }
    int n1 = 0, nhi = 0, nn = 0, jn = 0, stride = 0, isteps = 0, iprev = 0, iprevb = 0;
    double vlim = 0.0, cnext = 0.0, cnext1 = 0.0;
    // guard
    static constexpr double guard_rel = 3.0e-6;
    double cell_lo = 0.0, cell_hi = 0.0, vh0 = 0.0, vh1 = 0.0;
    double vsafe = 0.0; // scan steps entirely below this cannot involve the guard (the kernels' shortcuts for plain steps)
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
    double *xl, *yl;   // LDS Neville tables, element j at [j*XS] (j < NLO)
    // entries j >= NLO at [(j - NLO) * XH] (only if NLO < NEV_MAX); explicitly GLOBAL pointers: a generic pointer
    // would let the compiler fold the two cases into one flat access
    typedef __attribute__((address_space(1))) double gdouble;
    gdouble *xh = nullptr, *yh = nullptr;
    size_t XH = 0;
    __device__ __forceinline__ void set_high(double *xh_, double *yh_, size_t stride)
    {
        xh = (gdouble *)xh_;
        yh = (gdouble *)yh_;
        XH = stride;
    }
    __device__ __forceinline__ double nev_x(int j) const
    {
        if (NLO >= NEV_MAX || j < NLO) return xl[j * XS];
        return xh[(size_t)(j - NLO) * XH];
    }
    __device__ __forceinline__ double nev_y(int j) const
    {
        if (NLO >= NEV_MAX || j < NLO) return yl[j * XS];
        return yh[(size_t)(j - NLO) * XH];
    }
    __device__ __forceinline__ void nev_set_x(int j, double v)
    {
        if (NLO >= NEV_MAX || j < NLO) xl[j * XS] = v;
        else xh[(size_t)(j - NLO) * XH] = v;
    }
    __device__ __forceinline__ void nev_set_y(int j, double v)
    {
        if (NLO >= NEV_MAX || j < NLO) yl[j * XS] = v;
        else yh[(size_t)(j - NLO) * XH] = v;
    }
    double *vel;       // this model's output row (global)
    bool writer;       // this lane stores results (one lane per model)
    // state
    int k, root, st, ifirst, idir, nev, mnev, nctrl, errflag, iq, ift;
    bool active;
    double c1, c2, clow, del1, del2, del1st, c3, del3, ck, t1, omega, ceval;
    float t1a, t1b;
    unsigned int evals;
    // FAST only: the third point of the inverse-quadratic estimate (the bracket end last replaced), pass counter
    static constexpr double fast_tau = 5.0e-8;
    double cp = 0.0, delp = 0.0;
    bool have_p = false;
    int fit = 0;

    __device__ __forceinline__ void set_period(int kk)
    {
        const float h32 = 0.005f;
        double tt = per[kk];
        if (group || (!NOGROUP && has(F_CHAIN_A))) {
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
    // ifunc (1 Love, 2 Rayleigh) and counted (the counted scan is wanted) only matter for Love targets and the guard
    template <class MD>
    __device__ void init(const MD &md, int mmax, bool valid, int igr, int K_, const double *per_,
                         double *xl_, double *yl_, double *vel_, bool writer_, int mode_ = 1,
                         double *cper_ = nullptr, double *cbper_ = nullptr, int ifunc = 2, bool counted = false, bool refseq = false,
                         bool farith = false)
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
        group = !NOGROUP && igr == 1;
        put(F_CHAIN_A, !NOGROUP && igr == 2);
        flg &= ~F_ONE;
        K = K_;
        per = per_;
        xl = xl_;
        yl = yl_;
        vel = vel_;
        writer = writer_;
        mode = SIMPLE ? 1 : mode_;
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
        put(F_CNT_ON, counted && ifunc == 1);
        put(F_FA, FAST && farith);
        flg &= ~(F_CNT_OK | F_JUMP_READY);
        iprev = iprevb = 0;
        vlim = fmin(md.Bv(mmax - 1), betmxd);
        put(F_GUARD_ON, FAST && (PHASE_ONLY || (!group && !has(F_CHAIN_A) && !refseq)));
        // Rayleigh only: a Love scan may be the counted one, whose visited points differ -- the result must not depend on the scan mode
        put(F_SEED3, FAST && ifunc == 2 && has(F_GUARD_ON));
        flg &= ~F_GUARD;
        vh0 = md.Bv(mmax - 1);                          // half-space S velocity
        vh1 = (ifunc == 2) ? md.A(mmax - 1) : betmxd;   // half-space P velocity (Rayleigh: it enters |k - k_alpha| there)
        vsafe = has(F_GUARD_ON) ? fmin(vh0, vh1) : 1.0e300;
        if (active) set_period(0);
        ceval = c1;
        if (counted) plan_first_jump();
    }

    // The second root of the group velocity of period kk alone (:282-287), after init() with igr = 1 and mode 1: the first root is
    // read from the output row (a launch with igr = 2 put it there; 0 = that chain ended before period kk: nothing to do),
    // d1st = the value getsol keeps from the first search of the mode (`del1st`, :420: only its sign is used, :425-435).
    template <bool CNT = true>
    __device__ __forceinline__ void enter_second(int kk, double d1st)
    {
        if (NOGROUP || !active) return;
        const double ca = vel[kk];
        if (!(ca > 0.0)) {
            active = false;
            return;
        }
        flg |= F_ONE;
        k = kk;
        set_period(kk);
        ck = ca;
        root = 1;
        t1 = (double)t1b;
        omega = twopi / t1;
        ifirst = 0;
        del1st = d1st;
        clow = 0.0 + one * dc;
        c1 = ca - onea * dc;
        st = ST_FIRST;
        ceval = c1;
        if (CNT) plan_first_jump();
    }

    // The grid point `want` steps above `from` by the reference's repeated additions, stopping below vlim.
    // Returns the steps taken; *below = the grid point one step before the last.
    __device__ __forceinline__ int grid_ahead(double from, int want, double &to, double &below) const
    {
        int sdone = 0;
        double cs = from, cb = from;
        while (sdone < want) {
            const double nx = cs + dc;
            if (!(nx < vlim)) break;
            cb = cs;
            cs = nx;
            ++sdone;
        }
        to = cs;
        below = cb;
        return sdone;
    }
    // the first jump of a period's scan, worked out when the period is set up (the look-ahead evaluates it beside the
    // start value): valid if the start value's count turns out usable and the scan goes upward
    __device__ __forceinline__ void plan_first_jump()
    {
        flg &= ~F_JUMP_READY;
        if (!has(F_CNT_ON) || !active) return;
        const int ip = (!NOGROUP && root == 1) ? iprevb : iprev;
        stride = (ip > 0) ? ip + 1 : stride_first;
        jn = grid_ahead(c1, stride, cnext, cnext1);
        flg |= F_JUMP_READY;
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
    template <bool CNT = true> // (CNT = false: none of the counted scan's state is touched -- the Rayleigh wavefronts' instantiation)
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
            if (CNT) iprev = iprevb = 0;
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
        if (CNT) plan_first_jump();
    }

    // Look-ahead: candidate 0 is the pending request; candidate r > 0 is the phase velocity the r-th
    // request from now will most probably be for -- further bracket steps while stepping (:437-449),
    // further halvings towards the side on which a straight line through the bracket ends puts the
    // root while refining (:600-660).  Purely a guess about which values will be asked for: a value
    // is only ever consumed by advance() if it was computed for exactly the (ceval, omega) requested.
    template <bool CNT = true> // CNT = false: the caller knows the counted scan is off for this search (Rayleigh wavefronts)
    __device__ __forceinline__ double candidate(int r) const
    {
        double q = ceval;
        if ((CNT || FAST) && st >= ST_JUMP) { // the counted scan (guard probes: nothing to foresee)
            // The requests after the pending one are grid points around it: after the jump (k = jn steps above c1) the index
            // search starts, most probably, a step or a few below the target (the stride aims one step beyond the last period's
            // bracket); inside the index search they are the neighbours of the pending point (a root right behind or before it),
            // or -- the pending point being right below c2 -- nevill's first midpoint of that bracket.  The kernels take a
            // trial's value whenever the search asks for exactly that velocity, in whatever order (consume, CNT).
            if (CNT && r >= 1) {
                int k = -1;
                if (st == ST_JUMP) k = jn - r;
                else if (st == ST_BIS) {
                    if (r == 1 && !FAST && jn == nn - 1) return 0.5 * (ceval + c2);
                    const int d = (r + 1) / 2;
                    k = (r & 1) ? jn + d : jn - d;
                    if (k >= nn) k = -1;
                }
                if (k >= 1) {
                    q = c1;
                    for (int i = 0; i < k; ++i) q = q + dc;
                }
            }
            return q;
        }
        if (CNT && st == ST_FIRST && has(F_JUMP_READY)) { // start value of a period: the first jump of its counted scan, then the steps below it
            if (r == 1) q = cnext;
            else if (r == 2) q = cnext1;
            else if (r >= 3 && jn - (r - 1) >= 1) {
                q = c1;
                for (int i = 0; i < jn - (r - 1); ++i) q = q + dc;
            }
            return q;
        }
        if (FAST && st >= ST_FX) { // the only request that can be foreseen: the second point of an acceptance pair
            if (st == ST_FP1 && r == 1) {
                const double tau = fast_tau * fabs(c3);
                q = (c2 > c1) ? c3 + tau : c3 - tau;
            }
            return q;
        }
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

    // the scan step c1 -> c2 showed no sign change (:455-460): move on; returns the next `todo` of advance()
    template <bool CNT>
    __device__ __forceinline__ int step_done()
    {
        if (FAST) { // (the point the scan leaves behind: third point of the short refinement's first estimate, F_SEED3)
            cp = c1;
            delp = del1;
            have_p = true;
        }
        c1 = c2;
        del1 = del2;
        if (CNT) isteps += 1;
        return (c1 < cm || c1 >= betmxd + dc) ? 2 : 1;
    }
    // (c1, c2) is a bracket: set up its refinement; returns the next `todo` of advance()
    template <bool CNT>
    __device__ __forceinline__ int bracketed()
    {
        if (CNT) {
            if (!NOGROUP && root == 1) iprevb = isteps;
            else iprev = isteps;
        }
        if (FAST && (PHASE_ONLY || has(F_GUARD_ON))) {
            cell_lo = fmin(c1, c2);
            cell_hi = fmax(c1, c2);
            put(F_FLO_NEG, signs_differ((c1 < c2) ? del1 : del2, 0.0));
            // With the scan's previous point (Rayleigh) the inverse-quadratic estimate through three points is available at
            // once -- its error is ~1e-8, below tau -- and the refinement starts with the acceptance pair: two evaluations
            // (one round with two trial lanes) instead of three in two rounds.
            have_p = have_p && has(F_SEED3);
            put(F_SEEDED, have_p);
            fit = 0;
            // A bracket that contains betmx or a half-space velocity can hold THREE sign changes (the root, its mirror image
            // above the half-space velocity -- the half-space term takes |k - k_beta| -- and the first of the unphysical ones
            // beyond); which of them nevill ends at depends on its whole sequence.  The short sequence does not try: the guard
            // fires (the model is run again with the reference's).  (Round 6: the half-space velocities as well -- on models
            // drawn from a sampler's prior, whose half-space need not be the fastest layer, one in 10^4 came back with another
            // root of such a cell than the reference's.  The trial-per-lane kernel counts the cell's sign changes instead.)
            if ((cell_hi > betmxd && cell_lo < betmxd) || (cell_hi > vh0 && cell_lo < vh0) || (cell_hi > vh1 && cell_lo < vh1)) {
                flg |= F_GUARD;
                return 0;
            }
            return 7;
        }
        c3 = 0.5 * (c1 + c2);
        ceval = c3;
        st = ST_NEV0;
        return 0;
    }

    // nv: the packed mode count of the evaluation (LoveCount::packed; -1 = none).  CNT = false: the caller knows the counted
    // scan is off for this search (Rayleigh wavefronts: none of its code, none of its exec-mask bookkeeping)
    template <bool CNT = true>
    __device__ __forceinline__ void advance(double del, int nv = -1)
    {
        ++evals;
        // `todo`: 0 nothing, 1 prepare next bracket step, 2 root search failed (iret = -1),
        // 3 refinement finished with c3, 4 nevill top-of-loop, 5 nevill post-bracket section,
        // 6 root found, 7 next estimate of the short refinement, 10 the counted scan's index search
        int todo = 0;
        // fast arithmetic: a scan or guard value whose sign the rounding error could decide (or a poisoned one) is not trusted --
        // the guard fires (inside the refinement small values are what is expected: they move the root by < 1e-9 relative)
        if (FAST && has(F_FA) && !(fabs(del) >= fa::SIGN_FLOOR) && !(st >= ST_FX && st <= ST_FP2 && del == del)) flg |= F_GUARD;
        if ((CNT || FAST) && st >= ST_JUMP) {
            if (CNT && st == ST_JUMP) { // the grid point jn steps above c1
                if (nv < 0 || nv < n1) { // no usable count (the value is not used): the reference's steps from c1 on
                    flg &= ~F_CNT_OK;
                    todo = 1;
                } else if (nv == n1) { // jn steps without a sign change
                    c1 = ceval;
                    del1 = del;
                    isteps += jn;
                    stride = (stride < stride_next) ? stride_next : 2 * stride;
                    if (stride > stride_max) stride = stride_max;
                    todo = 1;
                } else {
                    c2 = ceval;
                    del2 = del;
                    nhi = nv;
                    nn = jn;
                    todo = 10;
                }
            } else if (CNT && st == ST_BIS) { // a grid point between c1 and c2
                if (nv < n1 || nv > nhi) { // (includes nv < 0)
                    flg &= ~F_CNT_OK;
                    todo = 1;
                } else if (nv > n1) {
                    c2 = ceval;
                    del2 = del;
                    nhi = nv;
                    nn = jn;
                    todo = 10;
                } else {
                    c1 = ceval;
                    del1 = del;
                    isteps += jn;
                    nn -= jn;
                    todo = 10;
                }
            } else if (FAST) { // guard probes
                if (st == ST_GH) {
                    if (signs_differ(del, 0.0) == has(F_FLO_NEG)) { // differs from the upper end's sign (the opposite of the lower end's)
                        flg |= F_GUARD;
                        todo = 3;
                    } else {
                        ceval = cell_lo - guard_rel * fabs(c3);
                        st = ST_GL;
                    }
                } else if (st == ST_GL) {
                    if (signs_differ(del, 0.0) != has(F_FLO_NEG)) flg |= F_GUARD;
                    todo = 3;
                } else if (st == ST_GS1) {
                    if (signs_differ(del, del1)) {
                        flg |= F_GUARD;
                    } else {
                        ceval = fmax(c1, c2) - guard_rel * fmax(c1, c2);
                        st = ST_GS2;
                    }
                } else {
                    if (signs_differ(del, del1)) flg |= F_GUARD;
                    else todo = step_done<CNT>();
                }
            }
            if (CNT && todo == 10) { // the index search between c1 (count n1) and c2 (nn steps above, count nhi > n1)
                if (nn > 1) {
                    int hh = nn / 2;
                    if (nhi - n1 == 1) { // exactly one sign change in between: a straight line through the two values
                        const double a1 = fabs(del1), a2 = fabs(del2);
                        const double tt = (double)nn * (a1 / (a1 + a2));
                        hh = (tt >= 1.0) ? ((tt < (double)(nn - 1)) ? (int)tt : nn - 1) : 1;
                    }
                    double cm_ = c1;
                    for (int i = 0; i < hh; ++i) cm_ = cm_ + dc;
                    jn = hh;
                    ceval = cm_;
                    st = ST_BIS;
                    todo = 0;
                } else if (!signs_differ(del1, del2)) { // an even number of roots inside one step: walked over
                    c1 = c2;
                    del1 = del2;
                    n1 = nhi;
                    isteps += 1;
                    stride = stride_next;
                    todo = 1;
                } else {
                    todo = bracketed<CNT>();
                }
            }
        } else
        if (FAST && st >= ST_FX) { // the short refinement's own states (kept out of the switch: the reference build's text stays as it was)
            if (st == ST_FX) {
                if (signs_differ(del, del1)) {
                    cp = c2; delp = del2;
                    c2 = ceval; del2 = del;
                } else {
                    cp = c1; delp = del1;
                    c1 = ceval; del1 = del;
                }
                have_p = true;
                todo = 7;
            } else if (st == ST_FP1) {
                if (signs_differ(del, del1)) { // the root is on the c1 side of x - tau
                    cp = c2; delp = del2;
                    c2 = ceval; del2 = del;
                    todo = 7;
                } else {
                    const double tau = fast_tau * fabs(c3);
                    const double x2 = (c2 > c1) ? c3 + tau : c3 - tau; // (the expression of candidate(1))
                    cp = c1; delp = del1;
                    c1 = ceval; del1 = del;
                    ceval = x2;
                    st = ST_FP2;
                }
                have_p = true;
            } else {
                if (signs_differ(del, del1)) {
                    todo = 3; // sign change inside [x - tau, x + tau]: c3 = x is the root
                } else {      // the root is beyond x + tau (the third point stays)
                    c1 = ceval; del1 = del;
                    todo = 7;
                }
            }
        } else
        switch (PHASE_ONLY ? (st == ST_FIRST ? ST_FIRST : ST_STEP) : st) { // (PHASE_ONLY: nevill's states are never entered)
        case ST_FIRST:
            del1 = del;
            if (ifirst == 1) del1st = del1;
            idir = (ifirst != 1 && signs_differ(del1st, del1)) ? -1 : +1;
            if (FAST) have_p = false; // (a new scan: no point left behind yet)
            if (CNT) {
                isteps = 0;
                n1 = nv;
                // (c1 + dc <= clow: getsol first moves the start to clow -- at its loop top, also when searching upward: a higher
                //  mode whose previous root lies below the floor the previous mode sets -- which changes the grid: plain steps then)
                put(F_CNT_OK, has(F_CNT_ON) && nv >= 0 && idir > 0 && c1 + dc > clow);
            }
            todo = 1;
            break;
        case ST_STEP:
            del2 = del;
            if (signs_differ(del1, del2)) { // bracketed: enter nevill with (c1,c2,del1,del2)
                todo = bracketed<CNT>();
            } else {
                bool probing = false;
                if (FAST && fmax(c1, c2) >= vsafe && !has(F_GUARD)) { // a step over a half-space velocity that showed no sign change: the guard's probes
                    const double lo = fmin(c1, c2), hi = fmax(c1, c2);
                    if ((lo <= vh0 && vh0 <= hi) || (lo <= vh1 && vh1 <= hi)) {
                        ceval = lo + guard_rel * hi;
                        st = ST_GS1;
                        probing = true;
                    }
                }
                if (!probing) todo = step_done<CNT>();
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
        if (FAST && todo == 7) { // next estimate inside the bracket (c1, del1), (c2, del2)
            fit = fit + 1;
            const double w = c2 - c1;
            if (fit >= 100 || fabs(w) <= 2.0 * (fast_tau * fabs(c1))) {
                c3 = 0.5 * (c1 + c2);
                todo = 3;
            } else {
                const double lo = fmin(c1, c2), hi = fmax(c1, c2);
                double x = 0.0;
                bool ok = false;
                if (have_p) {
                    const double d12 = del1 - del2, d1p = del1 - delp, d2p = del2 - delp;
                    if (d12 != 0.0 && d1p != 0.0 && d2p != 0.0) {
                        const double q1 = c1 * del2 * delp / (d12 * d1p);
                        const double q2 = c2 * del1 * delp / (d12 * d2p);
                        const double q3 = cp * del1 * del2 / (d1p * d2p);
                        x = q1 - q2 + q3;
                        ok = (x > lo && x < hi);
                    }
                }
                if (!ok) {
                    x = c1 - del1 * (c2 - c1) / (del2 - del1);
                    if (!(x > lo && x < hi)) x = 0.5 * (c1 + c2);
                }
                if (fit > 6) x = 0.5 * (c1 + c2);
                const double tau = fast_tau * fabs(x);
                const bool up = c2 > c1;
                const double x1 = up ? x - tau : x + tau;
                const double x2 = up ? x + tau : x - tau;
                const bool single = ((fit == 1 && !has(F_SEEDED)) || fit > 6 || !(x1 > lo && x1 < hi && x2 > lo && x2 < hi));
                c3 = x;
                ceval = single ? x : x1;
                st = single ? ST_FX : ST_FP1;
                todo = 0;
            }
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
                        nev_set_x(mnev, c3);
                        nev_set_y(mnev, del3);
                    } else {
                        xl[0] = c1;
                        yl[0] = del1;
                        xl[XS] = c2;
                        yl[XS] = del2;
                        mnev = 1;
                    }
                    const double ym = nev_y(mnev);
                    for (int kk = 1; kk <= mnev; ++kk) {
                        const int j = mnev - kk;
                        const double yj = nev_y(j);
                        const double denom = ym - yj;
                        if (fabs(denom) < 1.0e-10 * fabs(ym)) {
                            halve = true;
                            break;
                        }
                        nev_set_x(j, (-yj * nev_x(j + 1) + ym * nev_x(j)) / denom);
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
        if (FAST && todo == 3 && (flg & (F_GUARD_ON | F_GUARD)) == F_GUARD_ON && st < ST_GH) { // the guard at an accepted bracket (see the struct's comment)
            const double cn = c3, m2 = 2.0 * dc;
            if (fabs(cn - vh0) < m2 || fabs(cn - vh1) < m2 || fabs(cn - betmxd) < m2) {
                const double eps = guard_rel * fabs(cn);
                if (cell_hi - cn < eps || cn - cell_lo < eps || fabs(cn - betmxd) < eps) {
                    flg |= F_GUARD;
                } else {
                    ceval = cell_hi + eps;
                    st = ST_GH;
                    todo = 0;
                }
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
                        if (CNT) iprev = iprevb = 0;
                        next_search<CNT>();
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
                        if (CNT) plan_first_jump();
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
                if (!NOGROUP && has(F_CHAIN_A)) {
                    out = ck; // (unrounded: the launch of the second roots starts from it)
                } else if (!group) {
                    out = (double)cc0;
                } else { // all binary32 (:305)
                    const float cc1s = (float)c1b;
                    const float gvel =
                        (1.0f / t1a - 1.0f / t1b) / (1.0f / (t1a * cc0) - 1.0f / (t1b * cc1s));
                    out = (double)gvel;
                }
                if (writer) vel[k] = out;
                if (!NOGROUP && has(F_ONE)) {
                    active = false;
                } else {
                    k = k + 1;
                    next_search<CNT>();
                }
            }
            todo = 0;
        }
        if (CNT && todo == 1 && has(F_CNT_OK)) { // the counted scan: the grid point `stride` steps ahead (below vlim)
            if (!has(F_JUMP_READY)) jn = grid_ahead(c1, stride, cnext, cnext1);
            flg &= ~F_JUMP_READY;
            if (jn >= 2) {
                ceval = cnext;
                st = ST_JUMP;
                todo = 0;
            } else {
                flg &= ~F_CNT_OK; // too close to the limit: the reference's steps from here
            }
        }
        if (FAST && has(F_GUARD)) { // the guard fired: this run's results are not used (the engine runs the model again)
            active = false;
            return;
        }
        if (todo == 1) { // label 1000 of getsol: next bracket step (:437-446)
            if (CNT) flg &= ~F_JUMP_READY;
            c2 = (idir > 0) ? c1 + dc : c1 - dc;
            if (c2 <= clow) {
                idir = +1;
                c1 = clow;
                if (FAST) have_p = false; // (del1 is not the value at clow)
                c2 = c1 + dc;
                // dc > 0, so the retried c2 = clow + dc is above clow: no further loop
            }
            ceval = c2;
            st = ST_STEP;
        }
    }
};
using SearchRt = SearchT<0>;
using SearchRtFast = SearchT<0, NEV_MAX, 1>;

