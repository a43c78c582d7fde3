/*
 * oracle/csign_oracle.c -- CPU restatement of the engine's CERTIFIED-SIGN evaluation (csrc/swd_csign.h).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * NOT the reference's algorithm: the bracket scan of getsol (surfdisp96.f:437-460) consumes only the SIGN of the secular
 * function at its grid points.  The engine takes that sign, where it can prove it, from a CHEAP evaluation of the same
 * recursion: binary64 +, *, fma throughout (they cost what binary32 ones cost on this chip), but reciprocals and square
 * roots from a binary32 seed and two Newton steps, sin / cos / exp from short argument reductions and polynomials -- a
 * fifth of the instructions of the reference-exact evaluation (correctly rounded divisions, glibc-exact sincos / exp) --
 * carrying a running first-order error bound per vector component.  A grid point whose |value| does not exceed twice the
 * bound is evaluated with the reference-exact function as before.  This file restates the evaluation operation for
 * operation (IEEE arithmetic only, no libm call whose bits could differ between host and device): tests compare the
 * device's (value, bound) pairs bit for bit and hold the bound against the reference-exact recursion.
 *
 * The bound, in short (u = 2^-53; every statement first order in u, the final test carries a factor 2):
 *   - ra^2 = (k + k_a)|k - k_a| is formed as (om / (a c))^2 (a + c)|a - c|: no cancellation, relative error <= 16 u;
 *     p = ra d <= 10 u relative.
 *   - sin / cos of a propagating layer: |d cos| <= s, |d sin| <= s min(1, p), s = (10 max(p, 1) + 6) u (argument error
 *     10 u p, reduction and polynomial 6 u); evanescent layer, fac = exp(-2p): the same with s = (20 max(p, 1) + 8 +
 *     2 / min(p, 1)) u (the 1 - fac cancellation at small p).
 *   - every eigenfunction product T (cpcq, cpy, ... wz, a0pq) then has |dT| <= lam * That with lam = s_p + s_q + 16 u and
 *     the envelopes That built from |cos| <= 1, |w| <= W = min(1/ra, d), |x| <= X = ra min(1, p) (same for y, z).
 *   - every compound-matrix entry is a sum of coefficient x product; with the coefficients' magnitudes taken as
 *     gam + 1 for gam - 1 (its cancellation at gam = 1) the entry's error is <= (lam + 70 u) * M, M = the same formula
 *     with all terms positive on the envelopes.
 *   - one layer: ee_i = sum_j e_j ca_ji;  err_i = sum_j (eps_j |ca_ji| + |e_j| (lam + 70 u) M_ji); both divided by
 *     t = max |ee_i| (the reference's normc; a positive scale does not change the sign, so the reference-exact vector is
 *     thought of as scaled by the same t).  The reference-exact recursion's own rounding (correctly rounded operations,
 *     u / 2 each) is covered by the same terms.
 *   - the value is certified when |e_1| > 2 eps_1 at the surface.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "oracle.h"

#define U64 1.1102230246251565e-16 /* 2^-53 */

static inline double as_f64(uint64_t b)
{
    double f;
    memcpy(&f, &b, 8);
    return f;
}

/* 1 / x from the binary32 quotient and two Newton steps (x within binary32's range) */
static inline double rcp_fast(double x)
{
    const double y0 = (double)(1.0f / (float)x);
    const double y1 = fma(y0, fma(-x, y0, 1.0), y0);
    return fma(y1, fma(-x, y1, 1.0), y1);
}

/* 1 / sqrt(x) the same way; sqrt(x) = x * rsqrt(x) */
static inline double rsqrt_fast(double x)
{
    const double y0 = (double)(1.0f / sqrtf((float)x));
    const double y1 = fma(0.5 * y0, fma(-x * y0, y0, 1.0), y0);
    return fma(0.5 * y1, fma(-x * y1, y1, 1.0), y1);
}

/* sin and cos of 0 <= x < 1e5: two-part Cody-Waite reduction by pi/2, fdlibm's kernel polynomials on [-pi/4, pi/4] */
static void sincos_fast(double x, double *sn, double *cs)
{
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = fma(-n, 1.57079632673412561417e+00, x);
    r = fma(-n, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)n & 3;
    const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

/* exp(-x) for 0 <= x <= 64: |r| <= ln2 / 2, Taylor to r^12 */
static double expneg_fast(double x)
{
    const double t = -x;
    const double n = rint(t * 1.44269504088896338700e+00);
    double r = fma(-n, 6.93147180369123816490e-01, t);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = fma(r, 2.08767569878680989792e-09, 2.50521083854417187751e-08);
    p = fma(r, p, 2.75573192239858906526e-07);
    p = fma(r, p, 2.75573192239858906526e-06);
    p = fma(r, p, 2.48015873015873015873e-05);
    p = fma(r, p, 1.98412698412698412698e-04);
    p = fma(r, p, 1.38888888888888888889e-03);
    p = fma(r, p, 8.33333333333333333333e-03);
    p = fma(r, p, 4.16666666666666666667e-02);
    p = fma(r, p, 1.66666666666666666667e-01);
    p = fma(r, p, 0.5);
    const double e = fma(r * r, p, r) + 1.0;
    return e * as_f64((uint64_t)((int64_t)n + 1023) << 52);
}

/* One wave type of one layer: vel = the layer's velocity, c = the trial velocity, oc = om / c: cos-like, sin-like / r,
 * -+ r sin-like (the signs of surfdisp96.f:906-935), the evanescent exponent, and the envelopes and error level described
 * in the header. */
typedef struct {
    double cs, w, x, ex; /* values */
    double W, X, s;      /* envelopes of w and x, error level */
} wave_t;

static void wave_terms(double vel, double ivel, double c, double oc, double dpth, double idpth, wave_t *o)
{
    const double sa = vel - c;
    const double ia = oc * ivel;
    const double vc = vel + c;
    const double r2 = (ia * ia) * (vc * fabs(sa));
    const double rr = rsqrt_fast(r2);
    const double r = r2 * rr;
    const double p = r * dpth;
    const double pm = fmax(p, 1.0), pn = fmin(p, 1.0);
    /* the REFERENCE forms k - k_a by subtraction: its r carries the relative error u (a + c) / |a - c| of that cancellation
     * ((a + c) / |a - c| = ((a + c) ia)^2 / r2, without another division) */
    const double via = vc * ia;
    const double cn = (2.0 * U64) * ((via * via) * (rr * rr)) * (pm + 1.0);
    double sn, cs, s, ex = 0.0;
    if (sa < 0.0) { /* c above the layer velocity: propagating */
        sincos_fast(fmin(p, 9.0e4), &sn, &cs);
        o->x = -(r * sn);
        s = fma(10.0, pm, 6.0) * U64 + cn;
    } else {
        const double fac = (p < 16.0) ? expneg_fast(2.0 * p) : 0.0;
        cs = (1.0 + fac) * 0.5;
        sn = (1.0 - fac) * 0.5;
        o->x = r * sn;
        ex = p;
        s = (fma(20.0, pm, 8.0) + 2.0 * fmax(1.0, rr * idpth)) * U64 + cn; /* (2 / min(p, 1), 1 / p = (1 / r)(1 / d)) */
    }
    if (!(p < 9.0e4)) s = INFINITY; /* (beyond the reduction's range: never certified) */
    o->cs = cs;
    o->w = sn * rr;
    o->ex = ex;
    o->W = fmin(rr, dpth);
    o->X = r * pn;
    o->s = s;
}

/* Rayleigh: e = the surface vector under the per-layer max-norm scaling, eps = its error bounds.  Returns 1 when the sign
 * of e[0] is certified. */
static int rayleigh_cs(double omega, double c, const float *d, const float *a, const float *b, const float *rho,
                       int mmax, int llw, double *ev, double *epsv)
{
    for (int i = 0; i < 5; ++i) {
        ev[i] = 0.0;
        epsv[i] = INFINITY;
    }
    if (llw != 1 || mmax < 2) return 0; /* (water layer: not certified) */
    double om = omega;
    if (om < 1.0e-4) om = 1.0e-4;
    const double oc = om * rcp_fast(c), k2 = oc * oc, iom = rcp_fast(om);
    double e[5], eps[5];
    { /* half-space vector (surfdisp96.f:800-808) */
        const double ah = (double)a[mmax - 1], bh = (double)b[mmax - 1], rh = (double)rho[mmax - 1];
        const double ia = oc * rcp_fast(ah), ib = oc * rcp_fast(bh);
        const double ra2 = (ia * ia) * ((ah + c) * fabs(ah - c)), rb2 = (ib * ib) * ((bh + c) * fabs(bh - c));
        const double rsa = rsqrt_fast(ra2), rsb = rsqrt_fast(rb2);
        const double ra = ra2 * rsa, rb = rb2 * rsb;
        const double via = (ah + c) * ia, vib = (bh + c) * ib;
        const double t = bh * iom;
        const double gammk = 2.0 * t * t, gam = gammk * k2, gamm1 = gam - 1.0, g1 = gam + 1.0;
        const double rarb = ra * rb;
        e[0] = rh * rh * (gamm1 * gamm1 - gam * gammk * rarb);
        e[1] = -(rh * ra);
        e[2] = rh * (gamm1 - gammk * rarb);
        e[3] = rh * rb;
        e[4] = k2 - rarb;
        const double ku = 64.0 * U64; /* + the reference's cancellation in k - k_a, k - k_b (see wave_terms) */
        const double ka = ku + (2.0 * U64) * ((via * via) * (rsa * rsa)), kb = ku + (2.0 * U64) * ((vib * vib) * (rsb * rsb));
        eps[0] = rh * rh * (ku * (g1 * g1) + (ka + kb) * (gam * gammk * rarb));
        eps[1] = ka * (rh * ra);
        eps[2] = rh * (ku * g1 + (ka + kb) * (gammk * rarb));
        eps[3] = kb * (rh * rb);
        eps[4] = ku * k2 + (ka + kb) * rarb;
    }
    for (int m = mmax - 2; m >= 0; --m) {
        const double am = (double)a[m], bm = (double)b[m], rh = (double)rho[m], dm = (double)d[m];
        const double idm = rcp_fast(dm); /* (the device keeps 1 / a, 1 / b, 1 / rho, 1 / d of every layer in LDS: the same values) */
        wave_t P, Q;
        wave_terms(am, rcp_fast(am), c, oc, dm, idm, &P);
        wave_terms(bm, rcp_fast(bm), c, oc, dm, idm, &Q);
        const double t = bm * iom;
        const double gammk = 2.0 * t * t, gam = gammk * k2;
        const double exa = P.ex + Q.ex;
        const double a0 = (exa < 60.0) ? expneg_fast(exa) : 0.0;
        const double lam = P.s + Q.s + 86.0 * U64; /* products' 16 u + the coefficients' and sums' 70 u */
        /* eigenfunction products and their envelopes */
        const double cpcq = P.cs * Q.cs, cpy = P.cs * Q.w, cpz = P.cs * Q.x, cqw = Q.cs * P.w, cqx = Q.cs * P.x;
        const double xy = P.x * Q.w, xz = P.x * Q.x, wy = P.w * Q.w, wz = P.w * Q.x;
        const double Y = Q.W, Z = Q.X, W = P.W, X = P.X;
        const double XY = X * Y, XZ = X * Z, WY = W * Y, WZ = W * Z;
        /* coefficients (dnka, surfdisp96.f:1024-1068) and their magnitudes */
        const double gamm1 = gam - 1.0, twgm1 = gam + gamm1, gmgmk = gam * gammk, gmgm1 = gam * gamm1, gm1sq = gamm1 * gamm1;
        const double g1 = gam + 1.0, tw1 = gam + g1, gg1 = gam * g1, g1sq = g1 * g1;
        const double rho2 = rh * rh, ir = rcp_fast(rh), ir2 = ir * ir;
        const double a0pq = a0 - cpcq;
        const double k4 = k2 * k2;
        double ca[5][5], M[5][5];
        ca[0][0] = cpcq - 2.0 * gmgm1 * a0pq - gmgmk * xz - k2 * gm1sq * wy;
        ca[0][1] = (k2 * cpy - cqx) * ir;
        ca[0][2] = -(twgm1 * a0pq + gammk * xz + k2 * gamm1 * wy) * ir;
        ca[0][3] = (cpz - k2 * cqw) * ir;
        ca[0][4] = -(2.0 * k2 * a0pq + xz + k4 * wy) * ir2;
        ca[1][0] = (gmgmk * cpz - gm1sq * cqw) * rh;
        ca[1][1] = cpcq;
        ca[1][2] = gammk * cpz - gamm1 * cqw;
        ca[1][3] = -wz;
        ca[1][4] = ca[0][3];
        ca[3][0] = (gm1sq * cpy - gmgmk * cqx) * rh;
        ca[3][1] = -xy;
        ca[3][2] = gamm1 * cpy - gammk * cqx;
        ca[3][3] = cpcq;
        ca[3][4] = ca[0][1];
        ca[4][0] = -(2.0 * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * xz + gm1sq * gm1sq * wy) * rho2;
        ca[4][1] = ca[3][0];
        ca[4][2] = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * xz + gamm1 * gm1sq * wy) * rh;
        ca[4][3] = ca[1][0];
        ca[4][4] = ca[0][0];
        const double tt = -2.0 * k2;
        ca[2][0] = tt * ca[4][2];
        ca[2][1] = tt * ca[3][2];
        ca[2][2] = a0 + 2.0 * (cpcq - ca[0][0]);
        ca[2][3] = tt * ca[1][2];
        ca[2][4] = tt * ca[0][2];
        /* the same formulas on the envelopes (a0pq -> 2, cos -> 1), all terms positive */
        M[0][0] = 1.0 + 4.0 * gg1 + gmgmk * XZ + k2 * g1sq * WY;
        M[0][1] = (k2 * Y + X) * ir;
        M[0][2] = (2.0 * tw1 + gammk * XZ + k2 * g1 * WY) * ir;
        M[0][3] = (Z + k2 * W) * ir;
        M[0][4] = (4.0 * k2 + XZ + k4 * WY) * ir2;
        M[1][0] = (gmgmk * Z + g1sq * W) * rh;
        M[1][1] = 1.0;
        M[1][2] = gammk * Z + g1 * W;
        M[1][3] = WZ;
        M[1][4] = M[0][3];
        M[3][0] = (g1sq * Y + gmgmk * X) * rh;
        M[3][1] = XY;
        M[3][2] = g1 * Y + gammk * X;
        M[3][3] = 1.0;
        M[3][4] = M[0][1];
        M[4][0] = (4.0 * gmgmk * g1sq + gmgmk * gmgmk * XZ + g1sq * g1sq * WY) * rho2;
        M[4][1] = M[3][0];
        M[4][2] = (2.0 * gammk * g1 * tw1 + gam * gammk * gammk * XZ + g1 * g1sq * WY) * rh;
        M[4][3] = M[1][0];
        M[4][4] = M[0][0];
        const double t2 = 2.0 * k2;
        M[2][0] = t2 * M[4][2];
        M[2][1] = t2 * M[3][2];
        M[2][2] = 3.0 + 2.0 * M[0][0];
        M[2][3] = t2 * M[1][2];
        M[2][4] = t2 * M[0][2];
        double ee[5], er[5], t1 = 0.0;
        for (int i = 0; i < 5; ++i) {
            double acc = 0.0, err = 0.0, em = 0.0;
            for (int j = 0; j < 5; ++j) {
                acc = fma(e[j], ca[j][i], acc);
                err = fma(eps[j], fabs(ca[j][i]), err);
                em = fma(fabs(e[j]), M[j][i], em);
            }
            ee[i] = acc;
            er[i] = fma(lam, em, err);
            t1 = fmax(t1, fabs(acc));
        }
        if (!(t1 > 1.0e-30 && t1 < 1.0e30)) return 0;
        const double rt = rcp_fast(t1);
        for (int i = 0; i < 5; ++i) {
            e[i] = ee[i] * rt;
            eps[i] = fma(er[i], rt, 4.0 * U64 * fabs(e[i]));
        }
    }
    for (int i = 0; i < 5; ++i) {
        ev[i] = e[i];
        epsv[i] = eps[i];
    }
    return fabs(e[0]) > 2.0 * eps[0];
}

/* Love (surfdisp96.f:710-769): the 2-vector (e1, e2) at the surface and its bounds. */
static int love_cs(double omega, double c, const float *d, const float *b, const float *rho, int mmax, int llw,
                   double *ev, double *epsv)
{
    ev[0] = ev[1] = 0.0;
    epsv[0] = epsv[1] = INFINITY;
    if (llw != 1 || mmax < 2) return 0;
    const double oc = omega * rcp_fast(c);
    double e1, e2, p1, p2;
    {
        const double bh = (double)b[mmax - 1], rh = (double)rho[mmax - 1];
        const double ibh = rcp_fast(bh);
        const double ib = oc * ibh;
        const double rb2 = (ib * ib) * ((bh + c) * fabs(bh - c));
        const double rsb = rsqrt_fast(rb2);
        const double rb = rb2 * rsb;
        const double vib = (bh + c) * ib;
        e1 = rh * rb;
        e2 = ibh * ibh;
        p1 = (16.0 * U64 + (2.0 * U64) * ((vib * vib) * (rsb * rsb))) * e1; /* (the reference's cancellation in k - k_b) */
        p2 = 8.0 * U64 * e2;
    }
    for (int m = mmax - 2; m >= 0; --m) {
        const double bm = (double)b[m], rh = (double)rho[m], dm = (double)d[m];
        const double ibm = rcp_fast(bm);
        wave_t Q;
        wave_terms(bm, ibm, c, oc, dm, rcp_fast(dm), &Q);
        const double xmu = rh * bm * bm, ixmu = rcp_fast(rh) * ibm * ibm;
        const double lam = Q.s + 24.0 * U64;
        const double A = xmu * Q.x, Bq = Q.w * ixmu;
        const double MA = xmu * Q.X, MB = Q.W * ixmu;
        const double n1 = fma(e2, A, e1 * Q.cs);
        const double n2 = fma(e1, Bq, e2 * Q.cs);
        const double a1 = fabs(e1), a2 = fabs(e2), ac = fabs(Q.cs);
        const double r1 = fma(lam, fma(a2, MA, a1), fma(p2, fabs(A), p1 * ac));
        const double r2 = fma(lam, fma(a1, MB, a2), fma(p1, fabs(Bq), p2 * ac));
        const double t1 = fmax(fabs(n1), fabs(n2));
        if (!(t1 > 1.0e-30 && t1 < 1.0e30)) return 0;
        const double rt = rcp_fast(t1);
        e1 = n1 * rt;
        e2 = n2 * rt;
        p1 = fma(r1, rt, 4.0 * U64 * fabs(e1));
        p2 = fma(r2, rt, 4.0 * U64 * fabs(e2));
    }
    ev[0] = e1;
    ev[1] = e2;
    epsv[0] = p1;
    epsv[1] = p2;
    return fabs(e1) > 2.0 * p1;
}

/* the whole surface vector and its bounds (2 / 5 entries): for the test of the bound against the reference-exact recursion */
int bho_csign_vec(int ifunc, double omega, double c, const float *d, const float *a, const float *b, const float *rho,
                  int mmax, int llw, double *ev, double *epsv)
{
    if (ifunc == 1) return love_cs(omega, c, d, b, rho, mmax, llw, ev, epsv);
    return rayleigh_cs(omega, c, d, a, b, rho, mmax, llw, ev, epsv);
}

int bho_csign(int ifunc, double omega, double c, const float *d, const float *a, const float *b, const float *rho,
              int mmax, int llw, double *val, double *bound)
{
    double ev[5], epsv[5];
    const int ok = bho_csign_vec(ifunc, omega, c, d, a, b, rho, mmax, llw, ev, epsv);
    *val = ev[0];
    *bound = epsv[0];
    return ok;
}
