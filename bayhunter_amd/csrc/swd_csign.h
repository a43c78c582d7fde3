// bayhunter_amd/csrc/swd_csign.h -- the CERTIFIED-SIGN evaluation of the dispersion kernels (included by swd_common.h).
//
// getsol's bracket scan (surfdisp96.f:437-460) consumes only the SIGN of the secular function at its grid points.  Where it
// can be proven, the kernels take that sign from a cheap evaluation of the same recursion: binary64 +, *, fma throughout
// (they issue at the rate of binary32 ones on CDNA4), but reciprocals and square roots from a binary32 seed and two Newton
// steps, sin / cos / exp from short argument reductions and polynomials -- about a fifth of the instructions of the
// reference-exact evaluation (correctly rounded divisions, glibc-exact sincos / exp: bh_libm.h) -- run by ONE lane per
// grid point and carrying a running first-order error bound per vector component.  A grid point whose |value| does not
// exceed twice its bound is not certified: the scan evaluates it with the reference-exact function as before, so the
// brackets -- and every bit after them -- do not depend on this file.
//
// The bound (u = 2^-53; first order in u, the final test carries a factor 2):
//   - ra^2 = (k + k_a)|k - k_a| is formed as (om / (a c))^2 (a + c)|a - c|: no cancellation, <= 16 u relative; the
//     REFERENCE forms k - k_a by subtraction, its ra carries u (a + c) / |a - c| -- `cn` below;
//   - propagating layer: |d cos| <= s, |d sin| <= s min(1, p), s = (10 max(p, 1) + 6) u + cn; evanescent layer
//     (fac = exp(-2p)): the same with s = (20 max(p, 1) + 8 + 2 / min(p, 1)) u + cn (the 1 - fac cancellation at small p);
//   - every eigenfunction product T then has |dT| <= lam * That, lam = s_p + s_q + 16 u, the envelopes That built from
//     |cos| <= 1, |w| <= W = min(1/ra, d), |x| <= X = ra min(1, p) (same for y, z);
//   - every compound-matrix entry's error is <= (lam + 70 u) M, M = the entry's formula with all terms positive on the
//     envelopes and gam + 1 for |gam - 1|;
//   - one layer: ee_i = sum_j e_j ca_ji, err_i = sum_j (eps_j |ca_ji| + |e_j| (lam + 70 u) M_ji), both divided by
//     t = max |ee_i| (normc: a positive scale, the sign does not change).
// Every operation is an IEEE one (binary32 / and sqrt are correctly rounded in this build: hipcc's default), so
// oracle/csign_oracle.c restates the evaluation bit for bit on the CPU (tests/test_gpu_csign.py) and holds the bound against
// the reference-exact recursion (tests/test_oracle_csign.py).
#pragma once

namespace csign {
constexpr double U64 = 1.1102230246251565e-16; // 2^-53

__device__ __forceinline__ double rcp_fast(double x)
{
    const double y0 = (double)(1.0f / (float)x);
    const double y1 = __builtin_fma(y0, __builtin_fma(-x, y0, 1.0), y0);
    return __builtin_fma(y1, __builtin_fma(-x, y1, 1.0), y1);
}
__device__ __forceinline__ double rsqrt_fast(double x)
{
    const double y0 = (double)(1.0f / __builtin_sqrtf((float)x));
    const double y1 = __builtin_fma(0.5 * y0, __builtin_fma(-x * y0, y0, 1.0), y0);
    return __builtin_fma(0.5 * y1, __builtin_fma(-x * y1, y1, 1.0), y1);
}
// sin and cos of 0 <= x < 1e5: two-part Cody-Waite reduction by pi/2, fdlibm's kernel polynomials on [-pi/4, pi/4]
__device__ __forceinline__ void sincos_fast(double x, double &sn, double &cs)
{
    const double n = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-n, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-n, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double s = __builtin_fma(r * z, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double c = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)n & 3;
    const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    sn = (q & 2) ? -ss : ss;
    cs = ((q + 1) & 2) ? -cc : cc;
}
// exp(-x) for 0 <= x <= 64: |r| <= ln2 / 2, Taylor to r^12
__device__ __forceinline__ double expneg_fast(double x)
{
    const double t = -x;
    const double n = __builtin_rint(t * 1.44269504088896338700e+00);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, t);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    double p = __builtin_fma(r, 2.08767569878680989792e-09, 2.50521083854417187751e-08);
    p = __builtin_fma(r, p, 2.75573192239858906526e-07);
    p = __builtin_fma(r, p, 2.75573192239858906526e-06);
    p = __builtin_fma(r, p, 2.48015873015873015873e-05);
    p = __builtin_fma(r, p, 1.98412698412698412698e-04);
    p = __builtin_fma(r, p, 1.38888888888888888889e-03);
    p = __builtin_fma(r, p, 8.33333333333333333333e-03);
    p = __builtin_fma(r, p, 4.16666666666666666667e-02);
    p = __builtin_fma(r, p, 1.66666666666666666667e-01);
    p = __builtin_fma(r, p, 0.5);
    const double e = __builtin_fma(r * r, p, r) + 1.0;
    return e * __longlong_as_double((long long)((unsigned long long)((long long)n + 1023ll) << 52));
}

// One wave type of one layer (surfdisp96.f:906-935): cos-like, sin-like / r, -+ r sin-like, the evanescent exponent; the
// envelopes of w and x and the error level s (see the header of this file).
struct Wave {
    double cs, w, x, ex, W, X, s;
};
// ivel = 1 / vel, idpth = 1 / dpth as rcp_fast gives them (the group kernel keeps them in LDS per layer)
__device__ __forceinline__ void wave_terms(double vel, double ivel, double c, double oc, double dpth, double idpth, Wave &o)
{
    const double sa = vel - c;
    const double ia = oc * ivel;
    const double vc = vel + c;
    const double r2 = (ia * ia) * (vc * fabs(sa));
    const double rr = rsqrt_fast(r2);
    const double r = r2 * rr;
    const double p = r * dpth;
    const double pm = fmax(p, 1.0), pn = fmin(p, 1.0);
    const double via = vc * ia; // (a + c) / |a - c| = ((a + c) ia)^2 / r2
    const double cn = (2.0 * U64) * ((via * via) * (rr * rr)) * (pm + 1.0);
    double sn, cs, s, ex = 0.0;
    if (sa < 0.0) { // c above the layer velocity: propagating
        sincos_fast(fmin(p, 9.0e4), sn, cs);
        o.x = -(r * sn);
        s = __builtin_fma(10.0, pm, 6.0) * U64 + cn;
    } else {
        const double fac = (p < 16.0) ? expneg_fast(2.0 * p) : 0.0;
        cs = (1.0 + fac) * 0.5;
        sn = (1.0 - fac) * 0.5;
        o.x = r * sn;
        ex = p;
        s = (__builtin_fma(20.0, pm, 8.0) + 2.0 * fmax(1.0, rr * idpth)) * U64 + cn;
    }
    if (!(p < 9.0e4)) s = __builtin_inf(); // (beyond the reduction's range: never certified)
    o.cs = cs;
    o.w = sn * rr;
    o.ex = ex;
    o.W = fmin(rr, dpth);
    o.X = r * pn;
    o.s = s;
}

// Rayleigh.  Returns true when the sign of the surface value is certified; val / bound = e(1) under the per-layer
// max-norm scaling and its error bound (NaN / inf when the evaluation is unusable: never certified).
template <class MD>
__device__ __forceinline__ bool rayleigh(const MD &md, int mmax, int llw, double omega, double c, double &val, double &bound)
{
    val = 0.0;
    bound = __builtin_inf();
    if (llw != 1 || mmax < 2) return false; // (water layer: not certified)
    double om = omega;
    if (om < 1.0e-4) om = 1.0e-4;
    const double oc = om * rcp_fast(c), k2 = oc * oc, iom = rcp_fast(om);
    double e0, e1, e2, e3, e4, p0, p1, p2, p3, p4;
    { // half-space vector (surfdisp96.f:800-808)
        const double ah = md.A(mmax - 1), bh = md.Bv(mmax - 1), rh = md.R(mmax - 1);
        const double ia = oc * md.IA(mmax - 1), ib = oc * md.IB(mmax - 1);
        const double ra2 = (ia * ia) * ((ah + c) * fabs(ah - c)), rb2 = (ib * ib) * ((bh + c) * fabs(bh - c));
        const double rsa = rsqrt_fast(ra2), rsb = rsqrt_fast(rb2);
        const double ra = ra2 * rsa, rb = rb2 * rsb;
        const double via = (ah + c) * ia, vib = (bh + c) * ib;
        const double t = bh * iom;
        const double gammk = 2.0 * t * t, gam = gammk * k2, gamm1 = gam - 1.0, g1 = gam + 1.0;
        const double rarb = ra * rb;
        e0 = rh * rh * (gamm1 * gamm1 - gam * gammk * rarb);
        e1 = -(rh * ra);
        e2 = rh * (gamm1 - gammk * rarb);
        e3 = rh * rb;
        e4 = k2 - rarb;
        const double ku = 64.0 * U64;
        const double ka = ku + (2.0 * U64) * ((via * via) * (rsa * rsa)), kb = ku + (2.0 * U64) * ((vib * vib) * (rsb * rsb));
        p0 = rh * rh * (ku * (g1 * g1) + (ka + kb) * (gam * gammk * rarb));
        p1 = ka * (rh * ra);
        p2 = rh * (ku * g1 + (ka + kb) * (gammk * rarb));
        p3 = kb * (rh * rb);
        p4 = ku * k2 + (ka + kb) * rarb;
    }
    bool usable = true;
    for (int m = mmax - 2; m >= 0; --m) {
        const double am = md.A(m), bm = md.Bv(m), rh = md.R(m), dm = md.D(m);
        const double idm = md.ID(m);
        Wave P, Q;
        wave_terms(am, md.IA(m), c, oc, dm, idm, P);
        wave_terms(bm, md.IB(m), c, oc, dm, idm, Q);
        const double t = bm * iom;
        const double gammk = 2.0 * t * t, gam = gammk * k2;
        const double exa = P.ex + Q.ex;
        const double a0 = (exa < 60.0) ? expneg_fast(exa) : 0.0;
        const double lam = P.s + Q.s + 86.0 * U64;
        const double cpcq = P.cs * Q.cs, cpy = P.cs * Q.w, cpz = P.cs * Q.x, cqw = Q.cs * P.w, cqx = Q.cs * P.x;
        const double xy = P.x * Q.w, xz = P.x * Q.x, wy = P.w * Q.w, wz = P.w * Q.x;
        const double Y = Q.W, Z = Q.X, W = P.W, X = P.X;
        const double XY = X * Y, XZ = X * Z, WY = W * Y, WZ = W * Z;
        const double gamm1 = gam - 1.0, twgm1 = gam + gamm1, gmgmk = gam * gammk, gmgm1 = gam * gamm1, gm1sq = gamm1 * gamm1;
        const double g1 = gam + 1.0, tw1 = gam + g1, gg1 = gam * g1, g1sq = g1 * g1;
        const double rho2 = rh * rh, ir = md.IR(m), ir2 = ir * ir;
        const double a0pq = a0 - cpcq;
        const double k4 = k2 * k2;
        // compound matrix (dnka, surfdisp96.f:1024-1068): ca[j][i] as cJI with J, I = 1..5
        const double c11 = cpcq - 2.0 * gmgm1 * a0pq - gmgmk * xz - k2 * gm1sq * wy;
        const double c12 = (k2 * cpy - cqx) * ir;
        const double c13 = -(twgm1 * a0pq + gammk * xz + k2 * gamm1 * wy) * ir;
        const double c14 = (cpz - k2 * cqw) * ir;
        const double c15 = -(2.0 * k2 * a0pq + xz + k4 * wy) * ir2;
        const double c21 = (gmgmk * cpz - gm1sq * cqw) * rh;
        const double c22 = cpcq;
        const double c23 = gammk * cpz - gamm1 * cqw;
        const double c24 = -wz;
        const double c25 = c14;
        const double c41 = (gm1sq * cpy - gmgmk * cqx) * rh;
        const double c42 = -xy;
        const double c43 = gamm1 * cpy - gammk * cqx;
        const double c44 = cpcq;
        const double c45 = c12;
        const double c51 = -(2.0 * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * xz + gm1sq * gm1sq * wy) * rho2;
        const double c52 = c41;
        const double c53 = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * xz + gamm1 * gm1sq * wy) * rh;
        const double c54 = c21;
        const double c55 = c11;
        const double tt = -2.0 * k2;
        const double c31 = tt * c53;
        const double c32 = tt * c43;
        const double c33 = a0 + 2.0 * (cpcq - c11);
        const double c34 = tt * c23;
        const double c35 = tt * c13;
        // the same formulas on the envelopes (a0pq -> 2, cos -> 1), all terms positive
        const double M11 = 1.0 + 4.0 * gg1 + gmgmk * XZ + k2 * g1sq * WY;
        const double M12 = (k2 * Y + X) * ir;
        const double M13 = (2.0 * tw1 + gammk * XZ + k2 * g1 * WY) * ir;
        const double M14 = (Z + k2 * W) * ir;
        const double M15 = (4.0 * k2 + XZ + k4 * WY) * ir2;
        const double M21 = (gmgmk * Z + g1sq * W) * rh;
        const double M22 = 1.0;
        const double M23 = gammk * Z + g1 * W;
        const double M24 = WZ;
        const double M25 = M14;
        const double M41 = (g1sq * Y + gmgmk * X) * rh;
        const double M42 = XY;
        const double M43 = g1 * Y + gammk * X;
        const double M44 = 1.0;
        const double M45 = M12;
        const double M51 = (4.0 * gmgmk * g1sq + gmgmk * gmgmk * XZ + g1sq * g1sq * WY) * rho2;
        const double M52 = M41;
        const double M53 = (2.0 * gammk * g1 * tw1 + gam * gammk * gammk * XZ + g1 * g1sq * WY) * rh;
        const double M54 = M21;
        const double M55 = M11;
        const double t2 = 2.0 * k2;
        const double M31 = t2 * M53;
        const double M32 = t2 * M43;
        const double M33 = 3.0 + 2.0 * M11;
        const double M34 = t2 * M23;
        const double M35 = t2 * M13;
        const double a0e = fabs(e0), a1e = fabs(e1), a2e = fabs(e2), a3e = fabs(e3), a4e = fabs(e4);
#define BH_CS_COL(I, CA1, CA2, CA3, CA4, CA5, MM1, MM2, MM3, MM4, MM5)                                                     \
    double ee##I, er##I;                                                                                                   \
    {                                                                                                                      \
        double acc = 0.0, err = 0.0, em = 0.0;                                                                             \
        acc = __builtin_fma(e0, CA1, acc); err = __builtin_fma(p0, fabs(CA1), err); em = __builtin_fma(a0e, MM1, em);      \
        acc = __builtin_fma(e1, CA2, acc); err = __builtin_fma(p1, fabs(CA2), err); em = __builtin_fma(a1e, MM2, em);      \
        acc = __builtin_fma(e2, CA3, acc); err = __builtin_fma(p2, fabs(CA3), err); em = __builtin_fma(a2e, MM3, em);      \
        acc = __builtin_fma(e3, CA4, acc); err = __builtin_fma(p3, fabs(CA4), err); em = __builtin_fma(a3e, MM4, em);      \
        acc = __builtin_fma(e4, CA5, acc); err = __builtin_fma(p4, fabs(CA5), err); em = __builtin_fma(a4e, MM5, em);      \
        ee##I = acc;                                                                                                       \
        er##I = __builtin_fma(lam, em, err);                                                                               \
    }
        BH_CS_COL(0, c11, c21, c31, c41, c51, M11, M21, M31, M41, M51)
        BH_CS_COL(1, c12, c22, c32, c42, c52, M12, M22, M32, M42, M52)
        BH_CS_COL(2, c13, c23, c33, c43, c53, M13, M23, M33, M43, M53)
        BH_CS_COL(3, c14, c24, c34, c44, c54, M14, M24, M34, M44, M54)
        BH_CS_COL(4, c15, c25, c35, c45, c55, M15, M25, M35, M45, M55)
#undef BH_CS_COL
        double t1 = 0.0;
        t1 = fmax(t1, fabs(ee0));
        t1 = fmax(t1, fabs(ee1));
        t1 = fmax(t1, fabs(ee2));
        t1 = fmax(t1, fabs(ee3));
        t1 = fmax(t1, fabs(ee4));
        usable = usable && (t1 > 1.0e-30 && t1 < 1.0e30);
        const double rt = rcp_fast(t1);
        e0 = ee0 * rt; p0 = __builtin_fma(er0, rt, 4.0 * U64 * fabs(e0));
        e1 = ee1 * rt; p1 = __builtin_fma(er1, rt, 4.0 * U64 * fabs(e1));
        e2 = ee2 * rt; p2 = __builtin_fma(er2, rt, 4.0 * U64 * fabs(e2));
        e3 = ee3 * rt; p3 = __builtin_fma(er3, rt, 4.0 * U64 * fabs(e3));
        e4 = ee4 * rt; p4 = __builtin_fma(er4, rt, 4.0 * U64 * fabs(e4));
    }
    if (!usable) return false;
    val = e0;
    bound = p0;
    return fabs(e0) > 2.0 * p0;
}

// Love (surfdisp96.f:710-769)
template <class MD>
__device__ __forceinline__ bool love(const MD &md, int mmax, int llw, double omega, double c, double &val, double &bound)
{
    val = 0.0;
    bound = __builtin_inf();
    if (llw != 1 || mmax < 2) return false;
    const double oc = omega * rcp_fast(c);
    double e1, e2, p1, p2;
    {
        const double bh = md.Bv(mmax - 1), rh = md.R(mmax - 1);
        const double ibh = md.IB(mmax - 1);
        const double ib = oc * ibh;
        const double rb2 = (ib * ib) * ((bh + c) * fabs(bh - c));
        const double rsb = rsqrt_fast(rb2);
        const double rb = rb2 * rsb;
        const double vib = (bh + c) * ib;
        e1 = rh * rb;
        e2 = ibh * ibh;
        p1 = (16.0 * U64 + (2.0 * U64) * ((vib * vib) * (rsb * rsb))) * e1;
        p2 = 8.0 * U64 * e2;
    }
    bool usable = true;
    for (int m = mmax - 2; m >= 0; --m) {
        const double bm = md.Bv(m), rh = md.R(m), dm = md.D(m);
        const double ibm = md.IB(m);
        Wave Q;
        wave_terms(bm, ibm, c, oc, dm, md.ID(m), Q);
        const double xmu = rh * bm * bm, ixmu = md.IR(m) * ibm * ibm;
        const double lam = Q.s + 24.0 * U64;
        const double A = xmu * Q.x, Bq = Q.w * ixmu;
        const double MA = xmu * Q.X, MB = Q.W * ixmu;
        const double n1 = __builtin_fma(e2, A, e1 * Q.cs);
        const double n2 = __builtin_fma(e1, Bq, e2 * Q.cs);
        const double a1 = fabs(e1), a2 = fabs(e2), ac = fabs(Q.cs);
        const double r1 = __builtin_fma(lam, __builtin_fma(a2, MA, a1), __builtin_fma(p2, fabs(A), p1 * ac));
        const double r2 = __builtin_fma(lam, __builtin_fma(a1, MB, a2), __builtin_fma(p1, fabs(Bq), p2 * ac));
        const double t1 = fmax(fabs(n1), fabs(n2));
        usable = usable && (t1 > 1.0e-30 && t1 < 1.0e30);
        const double rt = rcp_fast(t1);
        e1 = n1 * rt;
        e2 = n2 * rt;
        p1 = __builtin_fma(r1, rt, 4.0 * U64 * fabs(e1));
        p2 = __builtin_fma(r2, rt, 4.0 * U64 * fabs(e2));
    }
    if (!usable) return false;
    val = e1;
    bound = p1;
    return fabs(e1) > 2.0 * p1;
}
} // namespace csign
