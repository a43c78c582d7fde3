/*
 * oracle/swd_oracle.c -- CPU restatement of the surf96 dispersion path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity: pinned (tests/test_oracle_swd.py).
 *
 * Restates src/extensions/surfdisp96.f of the reference (a Fortran-77 file) in C.  It is
 * written from the algorithm, with structured control flow, but it honours the things that
 * decide the bits of the result:
 *   - which quantities the Fortran keeps in binary32 (implicit typing in the driver,
 *     gtsolh entirely) and which in binary64 (the search, the secular functions),
 *   - binary32 literals that get widened (0.005, 1.5, 0.01 in nevill),
 *   - left-to-right evaluation of every product/sum (no FMA contraction, no re-association;
 *     compile with -ffp-contract=off),
 *   - the order of secular-function evaluations in the bracket search and in the hybrid
 *     bisection / inverse-Neville refinement, including which point is returned.
 */
#define _GNU_SOURCE /* sincos(): what the reference's compiled Fortran calls (not sin()+cos(), which are
                       different glibc implementations and differ in the last bit for ~0.1 % of arguments) */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle.h"

#define NLMAX 100 /* surfdisp96.f:59 */
#define NPMAX 60  /* surfdisp96.f:61 */

static _Thread_local int64_t g_neval;

static inline int signs_differ(double x, double y) { return (signbit(x) != 0) != (signbit(y) != 0); }

/* ---- Love: SH Thomson-Haskell, half-space up to the surface.  surfdisp96.f:710-769 ---- */
static double dltar1_vec(double wvno, double omega, const float *d, const float *b, const float *rho,
                         int mmax, int llw, double *vec)
{
    double beta1 = (double)b[mmax - 1];
    double rho1 = (double)rho[mmax - 1];
    double xkb = omega / beta1;
    double wvnop = wvno + xkb;
    double wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double e1 = rho1 * rb;
    double e2 = 1.0 / (beta1 * beta1);
    for (int m = mmax - 2; m >= llw - 1; --m) {
        double cosq, y, z;
        beta1 = (double)b[m];
        rho1 = (double)rho[m];
        double xmu = rho1 * beta1 * beta1;
        xkb = omega / beta1;
        wvnop = wvno + xkb;
        wvnom = fabs(wvno - xkb);
        rb = sqrt(wvnop * wvnom);
        double q = (double)d[m] * rb;
        if (wvno < xkb) { /* propagating */
            double sinq;
            sincos(q, &sinq, &cosq);
            y = sinq / rb;
            z = -rb * sinq;
        } else if (wvno == xkb) {
            cosq = 1.0;
            y = (double)d[m];
            z = 0.0;
        } else { /* evanescent, scaled by exp(-q) */
            double fac = 0.0;
            if (q < 16.0) fac = exp(-2.0 * q);
            cosq = (1.0 + fac) * 0.5;
            double sinq = (1.0 - fac) * 0.5;
            y = sinq / rb;
            z = rb * sinq;
        }
        double e10 = e1 * cosq + e2 * xmu * z;
        double e20 = e1 * y / xmu + e2 * cosq;
        double xnor = fabs(e10);
        double ynor = fabs(e20);
        if (ynor > xnor) xnor = ynor;
        if (xnor < 1.0e-40) xnor = 1.0;
        e1 = e10 / xnor;
        e2 = e20 / xnor;
    }
    if (vec) { /* (tests of the certified-sign evaluation: the whole surface vector) */
        vec[0] = e1;
        vec[1] = e2;
    }
    return e1;
}

double bho_dltar1(double wvno, double omega, const float *d, const float *b, const float *rho,
                  int mmax, int llw)
{
    return dltar1_vec(wvno, omega, d, b, rho, mmax, llw, NULL);
}

/* ---- Love: the same function with a MODE COUNT (not in the reference; restates swd_common.h, love count) -----------
 * The SH problem at fixed omega is a Sturm-Liouville problem in k^2: with (e1, e2) ~ (stress, displacement) of the solution
 * that decays in the half-space (valid for c < beta of the half-space), integrated upward as dltar1 does,
 *     N(c) = number of roots of dltar1(., omega) below c = Z + [e1 * e2 < 0 at the surface],
 * Z = zeros of the displacement in the finite layers (oscillation theorem: zeros enter at the free surface one by one, each
 * Dirichlet eigenvalue between two Neumann ones).  Zeros inside ONE layer: the displacement there is a pure sinusoid of
 * phase advance q = d * rb (or a cosh/sinh combination: at most one zero), so their number is floor(q / pi) or that + 1,
 * and the parity is the sign change of e2 across the layer -- which the recursion gives with the reference's own bits.
 * Hence sign(dltar1) == (-1)^N exactly as computed, and N(c2) - N(c1) certifies how many sign changes of the reference's
 * function lie between two trial velocities WITHOUT evaluating the grid points in between.
 * *valid = 0 where that argument does not hold: c >= beta of the half-space, or q / pi within 1e-9 of an integer in some
 * layer (floor ambiguous by two).  Returns the very bits of bho_dltar1. */
double bho_dltar1_count(double wvno, double omega, const float *d, const float *b, const float *rho,
                        int mmax, int llw, int *count, int *valid)
{
    const double rpi = 0.31830988618379067154; /* 1/pi */
    double beta1 = (double)b[mmax - 1];
    double rho1 = (double)rho[mmax - 1];
    double xkb = omega / beta1;
    double wvnop = wvno + xkb;
    double wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double e1 = rho1 * rb;
    double e2 = 1.0 / (beta1 * beta1);
    int n = 0, ok = (wvno > xkb);
    for (int m = mmax - 2; m >= llw - 1; --m) {
        double cosq, y, z;
        beta1 = (double)b[m];
        rho1 = (double)rho[m];
        double xmu = rho1 * beta1 * beta1;
        xkb = omega / beta1;
        wvnop = wvno + xkb;
        wvnom = fabs(wvno - xkb);
        rb = sqrt(wvnop * wvnom);
        double q = (double)d[m] * rb;
        int fl = 0;
        if (wvno < xkb) { /* propagating */
            double sinq;
            sincos(q, &sinq, &cosq);
            y = sinq / rb;
            z = -rb * sinq;
            const double x = q * rpi, xf = floor(x);
            if (!(x - xf > 1.0e-9 && xf + 1.0 - x > 1.0e-9) || !(x < 1.0e9)) ok = 0;
            fl = (x < 1.0e9) ? (int)xf : 0;
        } else if (wvno == xkb) {
            cosq = 1.0;
            y = (double)d[m];
            z = 0.0;
        } else { /* evanescent, scaled by exp(-q) */
            double fac = 0.0;
            if (q < 16.0) fac = exp(-2.0 * q);
            cosq = (1.0 + fac) * 0.5;
            double sinq = (1.0 - fac) * 0.5;
            y = sinq / rb;
            z = rb * sinq;
        }
        double e10 = e1 * cosq + e2 * xmu * z;
        double e20 = e1 * y / xmu + e2 * cosq;
        double xnor = fabs(e10);
        double ynor = fabs(e20);
        if (ynor > xnor) xnor = ynor;
        if (xnor < 1.0e-40) xnor = 1.0;
        const int flip = signs_differ(e20, e2);
        n += fl + (((fl & 1) != flip) ? 1 : 0);
        e1 = e10 / xnor;
        e2 = e20 / xnor;
    }
    *count = n + (signs_differ(e1, e2) ? 1 : 0);
    *valid = ok;
    return e1;
}

/* ---- Rayleigh helpers -------------------------------------------------------------- */

typedef struct {
    double a0, cpcq, cpy, cpz, cqw, cqx, xy, xz, wy, wz;
    double w, cosp; /* needed by the water-layer closure only */
} layer_terms;

/* Eigenfunction products with exponent bookkeeping.  surfdisp96.f:874-991 (`var`). */
static void layer_products(double p, double q, double ra, double rb, double wvno, double xka,
                           double xkb, double dpth, layer_terms *o)
{
    double cosp, cosq, w, x, y, z;
    double pex = 0.0, sex = 0.0;
    if (wvno < xka) {
        double sinp;
        sincos(p, &sinp, &cosp);
        w = sinp / ra;
        x = -ra * sinp;
    } else if (wvno == xka) {
        cosp = 1.0;
        w = dpth;
        x = 0.0;
    } else {
        pex = p;
        double fac = 0.0;
        if (p < 16.0) fac = exp(-2.0 * p);
        cosp = (1.0 + fac) * 0.5;
        double sinp = (1.0 - fac) * 0.5;
        w = sinp / ra;
        x = ra * sinp;
    }
    if (wvno < xkb) {
        double sinq;
        sincos(q, &sinq, &cosq);
        y = sinq / rb;
        z = -rb * sinq;
    } else if (wvno == xkb) {
        cosq = 1.0;
        y = dpth;
        z = 0.0;
    } else {
        sex = q;
        double fac = 0.0;
        if (q < 16.0) fac = exp(-2.0 * q);
        cosq = (1.0 + fac) * 0.5;
        double sinq = (1.0 - fac) * 0.5;
        y = sinq / rb;
        z = rb * sinq;
    }
    double exa = pex + sex;
    double a0 = 0.0;
    if (exa < 60.0) a0 = exp(-exa);
    o->a0 = a0;
    o->cpcq = cosp * cosq;
    o->cpy = cosp * y;
    o->cpz = cosp * z;
    o->cqw = cosq * w;
    o->cqx = cosq * x;
    o->xy = x * y;
    o->xz = x * z;
    o->wy = w * y;
    o->wz = w * z;
    o->w = w;
    o->cosp = cosp;
    /* The Fortran goes on to rescale cosq, y, z by exp(sex-pex) (:985-990); those values
     * are dead in the compound-matrix formulation (dnka only takes the products above), so
     * the exp() there has no observable effect and is not restated. */
}

/* Dunkin 5x5 compound matrix, surfdisp96.f:1024-1068 (`dnka`).  ca[j][i] == ca(j+1,i+1). */
static void compound_matrix(double ca[5][5], double wvno2, double gam, double gammk, double rho,
                            const layer_terms *v)
{
    const double two = 2.0;
    double gamm1 = gam - 1.0;
    double twgm1 = gam + gamm1;
    double gmgmk = gam * gammk;
    double gmgm1 = gam * gamm1;
    double gm1sq = gamm1 * gamm1;
    double rho2 = rho * rho;
    double a0pq = v->a0 - v->cpcq;
    ca[0][0] = v->cpcq - two * gmgm1 * a0pq - gmgmk * v->xz - wvno2 * gm1sq * v->wy;
    ca[0][1] = (wvno2 * v->cpy - v->cqx) / rho;
    ca[0][2] = -(twgm1 * a0pq + gammk * v->xz + wvno2 * gamm1 * v->wy) / rho;
    ca[0][3] = (v->cpz - wvno2 * v->cqw) / rho;
    ca[0][4] = -(two * wvno2 * a0pq + v->xz + wvno2 * wvno2 * v->wy) / rho2;
    ca[1][0] = (gmgmk * v->cpz - gm1sq * v->cqw) * rho;
    ca[1][1] = v->cpcq;
    ca[1][2] = gammk * v->cpz - gamm1 * v->cqw;
    ca[1][3] = -v->wz;
    ca[1][4] = ca[0][3];
    ca[3][0] = (gm1sq * v->cpy - gmgmk * v->cqx) * rho;
    ca[3][1] = -v->xy;
    ca[3][2] = gamm1 * v->cpy - gammk * v->cqx;
    ca[3][3] = ca[1][1];
    ca[3][4] = ca[0][1];
    ca[4][0] = -(two * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * v->xz + gm1sq * gm1sq * v->wy) * rho2;
    ca[4][1] = ca[3][0];
    ca[4][2] = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * v->xz + gamm1 * gm1sq * v->wy) * rho;
    ca[4][3] = ca[1][0];
    ca[4][4] = ca[0][0];
    double t = -two * wvno2;
    ca[2][0] = t * ca[4][2];
    ca[2][1] = t * ca[3][2];
    ca[2][2] = v->a0 + two * (v->cpcq - ca[0][0]);
    ca[2][3] = t * ca[1][2];
    ca[2][4] = t * ca[0][2];
}

/* ---- Rayleigh: Dunkin compound-matrix secular function.  surfdisp96.f:773-871 ---------- */
static double dltar4_vec(double wvno, double omga, const float *d, const float *a, const float *b,
                         const float *rho, int mmax, int llw, double *vec)
{
    double e[5], ee[5], ca[5][5];
    layer_terms v;
    double omega = omga;
    if (omega < 1.0e-4) omega = 1.0e-4;
    double wvno2 = wvno * wvno;
    double xka = omega / (double)a[mmax - 1];
    double xkb = omega / (double)b[mmax - 1];
    double wvnop = wvno + xka;
    double wvnom = fabs(wvno - xka);
    double ra = sqrt(wvnop * wvnom);
    wvnop = wvno + xkb;
    wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double t = (double)b[mmax - 1] / omega;
    /* E vector of the bottom half-space, :800-808 */
    double gammk = 2.0 * t * t;
    double gam = gammk * wvno2;
    double gamm1 = gam - 1.0;
    double rho1 = (double)rho[mmax - 1];
    e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
    e[1] = -rho1 * ra;
    e[2] = rho1 * (gamm1 - gammk * ra * rb);
    e[3] = rho1 * rb;
    e[4] = wvno2 - ra * rb;
    for (int m = mmax - 2; m >= llw - 1; --m) {
        xka = omega / (double)a[m];
        xkb = omega / (double)b[m];
        t = (double)b[m] / omega;
        gammk = 2.0 * t * t;
        gam = gammk * wvno2;
        wvnop = wvno + xka;
        wvnom = fabs(wvno - xka);
        ra = sqrt(wvnop * wvnom);
        wvnop = wvno + xkb;
        wvnom = fabs(wvno - xkb);
        rb = sqrt(wvnop * wvnom);
        double dpth = (double)d[m];
        rho1 = (double)rho[m];
        double p = ra * dpth;
        double q = rb * dpth;
        layer_products(p, q, ra, rb, wvno, xka, xkb, dpth, &v);
        compound_matrix(ca, wvno2, gam, gammk, rho1, &v);
        for (int i = 0; i < 5; ++i) {
            double cr = 0.0;
            for (int j = 0; j < 5; ++j) cr = cr + e[j] * ca[j][i];
            ee[i] = cr;
        }
        /* normc, :995-1020: max-norm rescale (the log of the norm is computed there and
         * never used). */
        double t1 = 0.0;
        for (int i = 0; i < 5; ++i)
            if (fabs(ee[i]) > t1) t1 = fabs(ee[i]);
        if (t1 < 1.0e-40) t1 = 1.0;
        for (int i = 0; i < 5; ++i) e[i] = ee[i] / t1;
    }
    if (vec)
        for (int i = 0; i < 5; ++i) vec[i] = e[i];
    if (llw != 1) { /* water layer on top, :850-866 (unreachable from BayHunter: vs > 0) */
        xka = omega / (double)a[0];
        wvnop = wvno + xka;
        wvnom = fabs(wvno - xka);
        ra = sqrt(wvnop * wvnom);
        double dpth = (double)d[0];
        rho1 = (double)rho[0];
        double p = ra * dpth;
        double znul = 1.0e-5;
        layer_products(p, znul, ra, znul, wvno, xka, znul, dpth, &v);
        double w0 = -rho1 * v.w;
        return v.cosp * e[0] + w0 * e[1];
    }
    return e[0];
}

double bho_dltar4(double wvno, double omga, const float *d, const float *a, const float *b,
                  const float *rho, int mmax, int llw)
{
    return dltar4_vec(wvno, omga, d, a, b, rho, mmax, llw, NULL);
}

/* The surface vector of the binary64 recursion (2 values for Love, 5 for Rayleigh): what tests/test_oracle_sign32.py holds the
 * certified-sign evaluation's error bound against. */
void bho_secular_vec(int ifunc, double omega, double c, const float *d, const float *a, const float *b, const float *rho,
                     int mmax, int llw, double *vec)
{
    if (ifunc == 1) dltar1_vec(omega / c, omega, d, b, rho, mmax, llw, vec);
    else dltar4_vec(omega / c, omega, d, a, b, rho, mmax, llw, vec);
}

typedef struct {
    const float *d, *a, *b, *rho;
    int mmax, llw, ifunc;
} medium;

static inline double secular(const medium *md, double wvno, double omega)
{
    ++g_neval;
    if (md->ifunc == 1) return bho_dltar1(wvno, omega, md->d, md->b, md->rho, md->mmax, md->llw);
    return bho_dltar4(wvno, omega, md->d, md->a, md->b, md->rho, md->mmax, md->llw);
}

/* Love only: value + mode count (bho_dltar1_count); *valid = 0 for Rayleigh (no such count for the P-SV problem here). */
static inline double secular_count(const medium *md, double wvno, double omega, int *count, int *valid)
{
    if (md->ifunc != 1) {
        *count = 0;
        *valid = 0;
        return secular(md, wvno, omega);
    }
    ++g_neval;
    return bho_dltar1_count(wvno, omega, md->d, md->b, md->rho, md->mmax, md->llw, count, valid);
}

/* ---- half-space Rayleigh velocity, 5 Newton steps, all binary32.  surfdisp96.f:367-388 -- */
float bho_gtsolh(float a, float b)
{
    float c = 0.95f * b;
    for (int i = 0; i < 5; ++i) {
        float gamma = b / a;
        float kappa = c / b;
        float k2 = kappa * kappa;
        float gk = gamma * kappa;
        float gk2 = gk * gk;
        float fac1 = sqrtf(1.0f - gk2);
        float fac2 = sqrtf(1.0f - k2);
        float tk = 2.0f - k2;
        float fr = tk * tk - 4.0f * fac1 * fac2;
        float frp = -4.0f * (2.0f - k2) * kappa + 4.0f * fac2 * gamma * gamma * kappa / fac1 +
                    4.0f * fac1 * kappa / fac2;
        frp = frp / b;
        c = c - fr / frp;
    }
    return c;
}

/* ---- root refinement: hybrid bisection / inverse Neville.  surfdisp96.f:557-686 -------- */
static double refine_root(const medium *md, double t, double c1, double c2, double del1, double del2)
{
    const double twopi = 2.0 * 3.141592653589793;
    const double omega = twopi / t;
    const double pct = (double)0.01f; /* `0.01*ss1` with a default-real literal, :623-626 */
    double x[20], y[20];
    int m = 1;
    int nev = 1; /* 0 force halving, 1 Neville allowed, 2 Neville running */
    double c3 = 0.5 * (c1 + c2);
    double del3 = secular(md, omega / c3, omega);
    for (int nctrl = 2; nctrl < 100; ++nctrl) {
        if (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)) { /* estimate left the bracket */
            nev = 0;
            c3 = 0.5 * (c1 + c2);
            del3 = secular(md, omega / c3, omega);
        }
        double s13 = del1 - del3;
        double s32 = del3 - del2;
        if (signs_differ(del3, del1)) {
            c2 = c3;
            del2 = del3;
        } else {
            c1 = c3;
            del1 = del3;
        }
        if (fabs(c1 - c2) <= 1.0e-6 * c1) break;
        if (signs_differ(s13, s32)) nev = 0;
        double ss1 = fabs(del1), s1 = pct * ss1;
        double ss2 = fabs(del2), s2 = pct * ss2;
        int halve = (s1 > ss2 || s2 > ss1 || nev == 0);
        if (!halve) {
            if (nev == 2) {
                x[m] = c3;
                y[m] = del3;
            } else {
                x[0] = c1;
                y[0] = del1;
                x[1] = c2;
                y[1] = del2;
                m = 1;
            }
            /* solve x(y=0) by Neville's scheme on the swapped table */
            for (int kk = 1; kk <= m; ++kk) {
                int j = m - kk; /* 0-based: Fortran j = m-kk+1 */
                double denom = y[m] - y[j];
                if (fabs(denom) < 1.0e-10 * fabs(y[m])) {
                    halve = 1;
                    break;
                }
                x[j] = (-y[j] * x[j + 1] + y[m] * x[j]) / denom;
            }
            if (!halve) {
                c3 = x[0];
                del3 = secular(md, omega / c3, omega);
                nev = 2;
                m = m + 1;
                if (m > 10) m = 10;
            }
        }
        if (halve) {
            c3 = 0.5 * (c1 + c2);
            del3 = secular(md, omega / c3, omega);
            nev = 1;
            m = 1;
        }
    }
    return c3; /* the last point evaluated, not a bracket end (:672) */
}

/* ---- the engine's short refinement (bh_engine_set_swd_search(e, BH_SEARCH_FAST), its default) -------------------
 * NOT the reference's algorithm: a CPU restatement of bayhunter_amd/csrc/swd_common.h (SearchT, FAST) so that the
 * device's sequence of evaluations in that mode can be checked bit for bit.  The gate of the mode itself is the
 * tolerance against the reference sequence above (north_star: 1e-5 relative on the velocities), see
 * tests/test_oracle_swd.py and tests/test_gpu_swd.py.
 * Same bracket as the reference (the scan in steps of dc is untouched); inside it: one regula-falsi point, then an
 * inverse-quadratic estimate x through the three known points, accepted when the function changes sign between
 * x - tau and x + tau (tau = 5e-8 |x|: the root is known twenty times closer than the reference's own stop test
 * |c1 - c2| <= 1e-6 c1 leaves it); a miss moves the bracket and repeats, bisection from the seventh pass on. */
#define BHO_FAST_TAU 5.0e-8
static int g_fast_search = 0; /* 0 reference sequence, 1 the short sequence as it is, 2 the short sequence with the guard below */
void bho_swd_set_search(int fast) { g_fast_search = (fast < 0 || fast > 2) ? 0 : fast; }
/* Scan mode (bh_engine_set_swd_scan): 0 = getsol's scan, one step of dc per evaluation; 1 (the engine's default) = the same
 * scan with the steps a MODE COUNT certifies as empty skipped (Love; bracket_and_refine).  Same brackets, same bits. */
static int g_scan_mode = 0;
void bho_swd_set_scan(int counted) { g_scan_mode = counted ? 1 : 0; }
static int g_stride_first = 16, g_stride_next = 4, g_stride_back = -1, g_stride_secant = 1; /* (tuning experiments) */
void bho_swd_set_scan_tuning(int first, int next, int back) { g_stride_first = first; g_stride_next = next % 100; g_stride_back = back; g_stride_secant = next < 100; }
/* Group velocities of the fundamental mode as the device runs them (bh_engine.hip launch_swd_jobs): first the chain
 * of the roots at t/(1+h) over all periods, then -- each on its own, in any order -- the roots at t/(1-h), which start from the
 * first root of their own period (:282-287) and from nothing else.  The same calls with the same arguments as one after the
 * other: tests/test_oracle_swd.py checks the bits against the reference's order (and, in this container, the compiled reference). */
static int g_group_split = 0;
void bho_swd_set_group_split(int on) { g_group_split = on ? 1 : 0; }
static int64_t g_guarded = 0; /* models the guard sent back to the reference sequence (statistics) */
int64_t bho_swd_guarded_count(int reset)
{
    int64_t v;
#pragma omp atomic read
    v = g_guarded;
    if (reset) {
#pragma omp atomic write
        g_guarded = 0;
    }
    return v;
}

static int g_fast_third = 1; /* (experiment) the scan's last replaced point seeds the inverse-quadratic estimate */
void bho_swd_set_fast_third(int on) { g_fast_third = on; }
static double refine_root_fast(const medium *md, double t, double c1, double c2, double del1, double del2, double betmx,
                               double cp0, double delp0, int have_p0)
{
    const double twopi = 2.0 * 3.141592653589793;
    const double omega = twopi / t;
    double cp = cp0, delp = delp0;
    int have_p = have_p0 && g_fast_third && md->ifunc == 2; /* (Rayleigh only: Love's scan may be the counted one, whose visited
                                                                points differ -- the result must not depend on the scan mode) */
    const int seeded = have_p;
    /* A bracket that reaches beyond the fastest S velocity (a root up there is rejected, :468-471, and the secular
     * function has further sign changes there): look at betmx first and keep the side below it if the root is there --
     * the one nevill walks into from its midpoint. */
    if (fmax(c1, c2) > betmx && fmin(c1, c2) < betmx) {
        const double fb = secular(md, omega / betmx, omega);
        const int low_is_1 = c1 < c2;
        const double flow = low_is_1 ? del1 : del2;
        if (signs_differ(fb, flow)) { /* sign change below betmx: betmx replaces the upper end */
            if (low_is_1) { cp = c2; delp = del2; c2 = betmx; del2 = fb; }
            else { cp = c1; delp = del1; c1 = betmx; del1 = fb; }
        } else {                      /* only above: betmx replaces the lower end (the search will fail) */
            if (low_is_1) { cp = c1; delp = del1; c1 = betmx; del1 = fb; }
            else { cp = c2; delp = del2; c2 = betmx; del2 = fb; }
        }
        have_p = 1;
    }
    for (int it = 1; it < 100; ++it) {
        const double w = c2 - c1;
        if (fabs(w) <= 2.0 * (BHO_FAST_TAU * fabs(c1))) break;
        const double lo = fmin(c1, c2), hi = fmax(c1, c2);
        double x = 0.0;
        int ok = 0;
        if (have_p) {
            const double d12 = del1 - del2, d1p = del1 - delp, d2p = del2 - delp;
            if (d12 != 0.0 && d1p != 0.0 && d2p != 0.0) {
                const double t1 = c1 * del2 * delp / (d12 * d1p);
                const double t2 = c2 * del1 * delp / (d12 * d2p);
                const double t3 = cp * del1 * del2 / (d1p * d2p);
                x = t1 - t2 + t3;
                ok = (x > lo && x < hi);
            }
        }
        if (!ok) {
            x = c1 - del1 * (c2 - c1) / (del2 - del1);
            if (!(x > lo && x < hi)) x = 0.5 * (c1 + c2);
        }
        if (it > 6) x = 0.5 * (c1 + c2);
        const double tau = BHO_FAST_TAU * fabs(x);
        const int up = c2 > c1;
        const double x1 = up ? x - tau : x + tau; /* towards c1 */
        const double x2 = up ? x + tau : x - tau; /* towards c2 */
        const int single = ((it == 1 && !seeded) || it > 6 || !(x1 > lo && x1 < hi && x2 > lo && x2 < hi));
        if (single) {
            const double fx = secular(md, omega / x, omega);
            if (signs_differ(fx, del1)) {
                cp = c2; delp = del2;
                c2 = x; del2 = fx;
            } else {
                cp = c1; delp = del1;
                c1 = x; del1 = fx;
            }
            have_p = 1;
            continue;
        }
        const double f1 = secular(md, omega / x1, omega);
        if (signs_differ(f1, del1)) { /* the root is on the c1 side of x1 */
            cp = c2; delp = del2;
            c2 = x1; del2 = f1;
            have_p = 1;
            continue;
        }
        cp = c1; delp = del1;
        c1 = x1; del1 = f1;
        have_p = 1;
        const double f2 = secular(md, omega / x2, omega);
        if (signs_differ(f2, del1)) return x; /* sign change inside [x1, x2] */
        c1 = x2; del1 = f2;                   /* the root is beyond x2 (the third point stays) */
    }
    return 0.5 * (c1 + c2);
}

/* ---- bracket search.  surfdisp96.f:390-482 (`getsol`).  Returns 1 ok / -1 failed. ------- */
/* The guard of search mode 2 (see bho_surfdisp96).  The half-space terms of both secular functions contain |k - k_v|
 * (surfdisp96.f:728-729, :793-797), so a root r creeping up to a half-space velocity v has a mirror-image sign change r'
 * just above v.  The reference's scan grid is anchored at the PREVIOUS root, which the short sequence knows to ~1e-6 only:
 * the two grids differ by s, |s| < 6e-6 km/s, and the two scans see different sign patterns exactly when a sign change lies
 * within |s| of a grid point.  With a lone root that only moves the bracket by a cell (same root); with the pair (r, r') it
 * decides whether the reference sees a sign change AT ALL.  So, with eps = 3e-6 x the velocity (|s| <= 1.05e-6 x it: the
 * reference stops at a bracket of 1e-6 c1, the short sequence at 5e-8):
 *   - a scan step [c1, c2] without a sign change that contains a half-space velocity may hide the pair: probe the secular
 *     function at eps inside both ends; a sign change there means the shifted grid could split the pair   -> guard;
 *   - an accepted bracket whose root lies within two steps of a half-space velocity (or of betmx): guard if the root is
 *     within eps of a bracket end, or if the function changes sign within eps OUTSIDE a bracket end (the image sitting
 *     right behind it), or if the root is within eps of betmx (the `c1 > betmx` test of getsol).
 * Probes cost evaluations only where a root is near a half-space velocity; the guard itself fires ~1e-3 of those times. */
#define BHO_GUARD_REL 3.0e-6 /* eps = this x the velocity: three times what the two sequences' roots can differ by */
#define BHO_GUARD_MARGIN (2.0 * (double)0.005f) /* "near": two scan steps (km/s) */
typedef struct {
    int on;       /* collect */
    int hit;      /* result */
    double vh[3]; /* half-space S, half-space P (Rayleigh), betmx */
    int nvh;
} guard_t;

static int guard_probe(const medium *md, double omega, double c, double ref)
{
    return signs_differ(secular(md, omega / c, omega), ref);
}

/* The counted scan (scan mode 1, Love).  getsol looks for the first step [g, g + dc] of its grid g_i = g_(i-1) + dc over
 * which the secular function changes sign.  With the mode count N (bho_dltar1_count; sign f == (-1)^N as computed, N the
 * number of sign changes below) the grid points need not all be visited: N(g_(i+s)) == N(g_i) proves that none of the s
 * steps in between shows a sign change -- the reference would walk through them -- and a larger count proves there is one, which a
 * bisection over the grid INDEX finds (the lowest step whose upper end counts more than g_i; taken as the bracket when the
 * difference is odd, walked over when even: two roots inside one step, invisible to the reference as well).  Every grid
 * point is formed by the reference's own repeated additions of dc, so the bracket handed to the refinement -- and every
 * bit after it -- is the reference's.  Upward scans below the slowest of (half-space S velocity, betmx) only; anything
 * else (downward scans, a scan that has turned round at clow, an ambiguous count) takes the reference's steps one by one.
 * The stride is a deterministic function of the search state: the first one aims one step beyond where the last
 * period's bracket was found (iprev + 1: g_stride_back = -1, as the device's plan_first_jump), then 4, 8, ... (first
 * period: 16, 32, 64). */
typedef struct {
    int on;
    int iprev; /* steps from the start value to the bracket of the previous period (0: none yet) */
} scan_t;

static void guard_after_root(const medium *md, double omega, double c1, double c2, double del1, double del2, double cn,
                             double betmx, guard_t *gd)
{
    const double lo = fmin(c1, c2), hi = fmax(c1, c2);
    const double flo = (c1 < c2) ? del1 : del2, fhi = (c1 < c2) ? del2 : del1;
    const double eps = BHO_GUARD_REL * fabs(cn);
    int near = 0;
    for (int i = 0; i < gd->nvh; ++i) near = near || fabs(cn - gd->vh[i]) < BHO_GUARD_MARGIN;
    if (!near) return;
    if (hi - cn < eps || cn - lo < eps || fabs(cn - betmx) < eps) gd->hit = 1;
    else if (guard_probe(md, omega, hi + eps, fhi)) gd->hit = 1;
    else if (guard_probe(md, omega, lo - eps, flo)) gd->hit = 1;
}

/* A bracketed root: the refinement (+ the guard of search mode 2).  Returns 1 ok / -1 failed / -2 the guard fired.
 * A bracket that contains betmx or a half-space velocity can hold THREE sign changes (the root, its mirror image above the
 * half-space velocity, and the first of the unphysical ones beyond): which of them nevill ends at depends on its whole
 * sequence, so the short sequence does not try -- the guard fires (SearchT::bracketed, swd_common.h: the same rule). */
static int refine_bracket(const medium *md, double t1, double omega, double c1, double c2, double del1, double del2,
                          double betmx, int fast, guard_t *gd, double *c1io, double cp, double delp, int have_p)
{
    if (fast && gd && gd->on) { /* betmx or a half-space velocity (gd->vh: S, P / betmx, betmx) inside the bracket */
        const double lo = fmin(c1, c2), hi = fmax(c1, c2);
        if ((hi > betmx && lo < betmx) || (hi > gd->vh[0] && lo < gd->vh[0]) || (hi > gd->vh[1] && lo < gd->vh[1])) {
            gd->hit = 1;
            return -2;
        }
    }
    const double cn = fast ? refine_root_fast(md, t1, c1, c2, del1, del2, betmx, cp, delp, have_p) : refine_root(md, t1, c1, c2, del1, del2);
    *c1io = cn;
    if (fast && gd && gd->on) {
        guard_after_root(md, omega, c1, c2, del1, del2, cn, betmx, gd);
        if (gd->hit) return -2;
    }
    if (cn > betmx) return -1;
    return 1;
}

static int bracket_and_refine(const medium *md, double t1, double *c1io, double clow, double dc,
                              double cm, double betmx, int ifirst, double *del1st, int fast, guard_t *gd, scan_t *sc)
{
    const double twopi = 2.0 * 3.141592653589793;
    double c1 = *c1io, c2;
    double omega = twopi / t1;
    int n1 = 0, v1 = 0;
    const int counted = sc && sc->on && md->ifunc == 1;
    double cp = 0.0, delp = 0.0;                        /* the evaluated point a bracket end last replaced (third point of the */
    int have_p = 0;                                     /* short refinement's first estimate) */
    double del1 = 0.0;
    del1 = counted ? secular_count(md, omega / c1, omega, &n1, &v1) : secular(md, omega / c1, omega);
    if (ifirst == 1) *del1st = del1;
    int idir = 1;
    if (ifirst != 1 && signs_differ(*del1st, del1)) idir = -1;
    /* (c1 + dc <= clow: getsol first moves the start to clow -- its loop top, also when searching upward: a higher mode whose
       previous root lies below the floor the previous mode sets -- and that changes the grid: the reference's steps then) */
    int use_count = counted && v1 && idir > 0 && c1 + dc > clow;
    int isteps = 0;                                     /* steps taken from the start value */
    int stride = 0;
    if (use_count) stride = (sc->iprev > 0) ? sc->iprev - g_stride_back : g_stride_first;
    const double vlim = fmin((double)md->b[md->mmax - 1], betmx);
    for (;;) {
        if (use_count) {
            int s = 0;
            double cs = c1;
            while (s < stride) {
                const double nx = cs + dc;
                if (!(nx < vlim)) break;
                cs = nx;
                ++s;
            }
            if (s >= 2) {
                int ns = 0, vs = 0;
                const double dels = secular_count(md, omega / cs, omega, &ns, &vs);
                if (!vs || ns < n1) {
                    use_count = 0; /* (the value is not used: the reference's steps from c1 on) */
                    continue;
                }
                if (ns == n1) { /* s steps without a sign change */
                    cp = c1; delp = del1; have_p = 1;
                    c1 = cs;
                    del1 = dels;
                    isteps += s;
                    stride = (stride < g_stride_next) ? g_stride_next : 2 * stride;
                    if (stride > 64) stride = 64;
                    continue;
                }
                /* the lowest step whose upper end counts more than n1 */
                double chi = cs, delhi = dels;
                int nhi = ns, n = s, ok = 1;
                while (n > 1) {
                    /* where to look next: a straight line through the two known values when exactly one sign change
                     * lies between them (regula falsi over the grid index), the middle otherwise */
                    int h = n / 2;
                    if (g_stride_secant && nhi - n1 == 1) {
                        const double a1 = fabs(del1), a2 = fabs(delhi);
                        const double tt = (double)n * (a1 / (a1 + a2));
                        h = (tt >= 1.0) ? ((tt < (double)(n - 1)) ? (int)tt : n - 1) : 1;
                    }
                    double cmid = c1;
                    for (int i = 0; i < h; ++i) cmid = cmid + dc;
                    int nm = 0, vm = 0;
                    const double dm = secular_count(md, omega / cmid, omega, &nm, &vm);
                    if (!vm || nm < n1 || nm > nhi) {
                        ok = 0;
                        break;
                    }
                    if (nm > n1) {
                        cp = chi; delp = delhi; have_p = 1;
                        chi = cmid; delhi = dm; nhi = nm;
                        n = h;
                    } else {
                        cp = c1; delp = del1; have_p = 1;
                        c1 = cmid; del1 = dm;
                        isteps += h;
                        n = n - h;
                    }
                }
                if (!ok) {
                    use_count = 0;
                    continue;
                }
                if (!signs_differ(del1, delhi)) { /* an even number of roots inside one step: walked over */
                    cp = c1; delp = del1; have_p = 1;
                    c1 = chi;
                    del1 = delhi;
                    n1 = nhi;
                    isteps += 1;
                    stride = g_stride_next;
                    continue;
                }
                c2 = chi; /* == c1 + dc */
                if (sc) sc->iprev = isteps;
                return refine_bracket(md, t1, omega, c1, c2, del1, delhi, betmx, fast, gd, c1io, cp, delp, have_p);
            }
            use_count = 0; /* too close to the limit: the reference's steps from here */
        }
        c2 = (idir > 0) ? c1 + dc : c1 - dc;
        if (c2 <= clow) { /* never search below clow: turn round, restart from clow */
            idir = 1;
            c1 = clow;
            have_p = 0; /* (del1 is not the value at clow: no third point, and see refine_root_fast) */
            continue;
        }
        omega = twopi / t1;
        double del2 = secular(md, omega / c2, omega);
        if (signs_differ(del1, del2)) {
            if (sc) sc->iprev = isteps;
            return refine_bracket(md, t1, omega, c1, c2, del1, del2, betmx, fast, gd, c1io, cp, delp, have_p);
        }
        if (fast && gd && gd->on && !gd->hit) { /* a step over a half-space velocity that showed no sign change */
            const double lo = fmin(c1, c2), hi = fmax(c1, c2);
            int over = 0;
            for (int i = 0; i < 2 && i < gd->nvh; ++i) over = over || (lo <= gd->vh[i] && gd->vh[i] <= hi);
            const double eps = BHO_GUARD_REL * hi;
            if (over && (guard_probe(md, omega, lo + eps, del1) || guard_probe(md, omega, hi - eps, del1))) {
                gd->hit = 1;
                return -2; /* the model is run again with the reference's sequence: nothing more to do here */
            }
        }
        cp = c1; delp = del1; have_p = 1;
        c1 = c2;
        del1 = del2;
        isteps += 1;
        if (c1 < cm) break;
        if (c1 >= betmx + dc) break;
    }
    *c1io = c1;
    return -1;
}

/* ---- earth flattening.  surfdisp96.f:486-553 (`sphere`) ------------------------------- */
typedef struct {
    float rtp[NLMAX], dtp[NLMAX], btp[NLMAX];
    float dhalf;
} sphere_state;

static float powi_f32(float a, int b) /* integer power the way compiler-rt's __powisf2 does it */
{
    int recip = b < 0;
    float r = 1.0f;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}

static void sphere_init(float *d, float *a, float *b, const float *rho, int mmax, sphere_state *s)
{
    const double ar = 6370.0;
    double dr = 0.0, r0 = ar;
    d[mmax - 1] = 1.0f;
    for (int i = 0; i < mmax; ++i) {
        s->dtp[i] = d[i];
        s->rtp[i] = rho[i];
    }
    for (int i = 0; i < mmax; ++i) {
        dr = dr + (double)d[i];
        double r1 = ar - dr;
        double z0 = ar * log(ar / r0);
        double z1 = ar * log(ar / r1);
        d[i] = (float)(z1 - z0);
        double tmp = (ar + ar) / (r0 + r1); /* layer mid-point */
        a[i] = (float)((double)a[i] * tmp);
        b[i] = (float)((double)b[i] * tmp);
        s->btp[i] = (float)tmp;
        r0 = r1;
    }
    s->dhalf = d[mmax - 1];
    d[mmax - 1] = 0.0f;
}

static void sphere_density(int ifunc, float *d, float *rho, int mmax, const sphere_state *s)
{
    d[mmax - 1] = s->dhalf;
    for (int i = 0; i < mmax; ++i) {
        if (ifunc == 1)
            rho[i] = s->rtp[i] * powi_f32(s->btp[i], -5);
        else
            rho[i] = s->rtp[i] * powf(s->btp[i], -2.275f);
    }
    d[mmax - 1] = 0.0f;
}

/* ---- driver.  surfdisp96.f:55-360 ----------------------------------------------------
 * fast_mode: 0 = the reference's sequence (the restatement proper); 1 = phase-velocity roots refined by the engine's short
 * sequence (refine_root_fast).  *guard (fast_mode 1 only; may be NULL) is set when the run met one of the situations in
 * which the REFERENCE's own outcome hinges on the last bits of a previous root -- see bho_surfdisp96 below. */
static int surfdisp96_run(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                          int nlayer, int iflsph, int iwave, int mode, int igr, int kmax,
                          const double *t, double *cg, int fast_mode, int *guard)
{
    float d[NLMAX], a[NLMAX], b[NLMAX], rho[NLMAX];
    double c[NPMAX], cb[NPMAX];
    sphere_state sph;
    int err = 0;
    const int mmax = nlayer;
    for (int i = 0; i < mmax; ++i) {
        b[i] = vsm[i];
        a[i] = vpm[i];
        d[i] = thkm[i];
        rho[i] = rhom[i];
    }
    const int ifunc = (iwave == 1) ? 1 : 2;
    const float ddc = 0.005f, sone = 1.5f, h = 0.005f;
    const int llw = (b[0] <= 0.0f) ? 2 : 1;
    const double one = 1.0e-2;
    if (iflsph == 1) sphere_init(d, a, b, rho, mmax, &sph);

    /* extremal velocities, :145-156 (binary32 compares) */
    float betmx = -1.e20f, betmn = 1.e20f;
    int jmn = 0, jsol = 1;
    for (int i = 0; i < mmax; ++i) {
        if (b[i] > 0.01f && b[i] < betmn) {
            betmn = b[i];
            jmn = i;
            jsol = 1;
        } else if (b[i] <= 0.01f && a[i] < betmn) {
            betmn = a[i];
            jmn = i;
            jsol = 0;
        }
        if (b[i] > betmx) betmx = b[i];
    }
    if (iflsph == 1) sphere_density(ifunc, d, rho, mmax, &sph);

    medium md = {d, a, b, rho, mmax, llw, ifunc};
    const double onea = (double)sone;
    /* start value: half-space Rayleigh velocity of the slowest layer, backed off twice */
    float cc1 = (jsol == 0) ? betmn : bho_gtsolh(a[jmn], b[jmn]);
    cc1 = 0.95f * cc1;
    cc1 = 0.90f * cc1;
    const double cc = (double)cc1;
    const double dc = fabs((double)ddc);
    double c1 = cc;
    const double cm = cc;
    for (int i = 0; i < kmax; ++i) {
        cb[i] = 0.0;
        c[i] = 0.0;
    }
    guard_t gd = {guard != NULL && fast_mode && igr == 0 && mode == 1, 0, {(double)b[mmax - 1], (double)betmx, (double)a[mmax - 1]}, ifunc == 2 ? 3 : 2};
    { /* (order: half-space S, then -- Rayleigh only, where it also enters |k - k_alpha| -- half-space P; betmx last) */
        gd.vh[1] = (ifunc == 2) ? (double)a[mmax - 1] : (double)betmx;
        gd.vh[2] = (double)betmx;
    }
    scan_t sc = {g_scan_mode, 0}, sc2 = {g_scan_mode, 0}; /* first / second root of a period */
    double del1st = 0.0; /* Fortran SAVE variable; always (re)set when ifirst == 1 */
    int ift = 999;       /* 1-based index of the first period a previous mode failed at */
    for (int iq = 1; iq <= mode; ++iq) {
        int k;
        int failed = 0;
        sc.iprev = sc2.iprev = 0;
        for (k = 1; k <= kmax; ++k) {
            if (k >= ift) {
                failed = 1;
                break;
            }
            double t1 = t[k - 1];
            float t1a, t1b = 0.0f;
            if (igr > 0) {
                t1a = (float)(t1 / (double)(1.0f + h));
                t1b = (float)(t1 / (double)(1.0f - h));
                t1 = (double)t1a;
            } else {
                t1a = (float)t1;
            }
            double clow;
            int ifirst;
            if (k == 1 && iq == 1) {
                c1 = cc;
                clow = cc;
                ifirst = 1;
            } else if (k == 1 && iq > 1) {
                c1 = c[0] + one * dc;
                clow = c1;
                ifirst = 1;
            } else if (k > 1 && iq > 1) {
                ifirst = 0;
                clow = c[k - 1] + one * dc;
                c1 = c[k - 2];
                if (c1 < clow) c1 = clow;
            } else {
                ifirst = 0;
                c1 = c[k - 2] - onea * dc;
                clow = cm;
            }
            /* (the short refinement applies to phase-velocity runs only: a group velocity is a difference quotient of
               two roots and amplifies their 1e-6 scatter a hundredfold) */
            /* ... and to the fundamental mode only (mode == 1, the reference's default): higher modes lie close together at
               short periods -- pairs of roots inside one scan step, which a 1e-6 shift of the grid splits or not, and no guard
               can see them without a mode count (found by tools/gpu_fuzz.py in round 4: a Love mode-2 target at T = 1 s) */
            const int fast = fast_mode && igr == 0 && mode == 1;
            int iret = bracket_and_refine(&md, t1, &c1, clow, dc, cm, (double)betmx, ifirst, &del1st, fast, &gd, &sc);
            if (iret == -2) { /* the guard fired: this run's results are not used */
                *guard = 1;
                return 0;
            }
            if (iret == -1) {
                failed = 1;
                break;
            }
            c[k - 1] = c1;
            if (igr > 0 && g_group_split && mode == 1) { /* (the second roots follow the chain of the first ones, below) */
                cg[k - 1] = 0.0;
                continue;
            }
            if (igr > 0) { /* second root at the slightly longer period */
                t1 = (double)t1b;
                clow = cb[k - 1] + one * dc;
                c1 = c1 - onea * dc;
                iret = bracket_and_refine(&md, t1, &c1, clow, dc, cm, (double)betmx, 0, &del1st, 0, NULL, &sc2);
                if (iret == -1) c1 = c[k - 1];
                cb[k - 1] = c1;
            } else {
                c1 = 0.0;
            }
            float cc0 = (float)c[k - 1];
            float cc1s = (float)c1;
            if (igr == 0) {
                cg[k - 1] = (double)cc0;
            } else { /* all binary32, :305 */
                float gvel = (1.0f / t1a - 1.0f / t1b) / (1.0f / (t1a * cc0) - 1.0f / (t1b * cc1s));
                cg[k - 1] = (double)gvel;
            }
        }
        if (failed) {
            if (iq == 1) err = 1;
            ift = k;
            for (int i = k; i <= kmax; ++i) cg[i - 1] = 0.0;
        }
        if (igr > 0 && g_group_split && mode == 1) {
            /* the second roots of the periods the chain reached, LAST period first (any order will do: nothing carries over
               from one to the next but the counted scan's stride, which starts anew -- same bracket, other evaluation count) */
            const int reached = failed ? k - 1 : kmax;
            for (int kk = reached; kk >= 1; --kk) {
                const float t1a = (float)(t[kk - 1] / (double)(1.0f + h)), t1b = (float)(t[kk - 1] / (double)(1.0f - h));
                scan_t sc3 = {g_scan_mode, 0};
                double c2nd = c[kk - 1] - onea * dc;
                const int iret = bracket_and_refine(&md, (double)t1b, &c2nd, 0.0 + one * dc, dc, cm, (double)betmx, 0, &del1st, 0, NULL, &sc3);
                if (iret == -1) c2nd = c[kk - 1];
                const float cc0 = (float)c[kk - 1], cc1s = (float)c2nd;
                cg[kk - 1] = (double)((1.0f / t1a - 1.0f / t1b) / (1.0f / (t1a * cc0) - 1.0f / (t1b * cc1s)));
            }
        }
    }
    return err;
}

/* Search mode 2 (bh_engine_set_swd_search(e, BH_SEARCH_FAST) as the engine runs it): the short sequence, GUARDED.  The
 * scan grid of a period is anchored at the previous period's root (c1 = c(k-1) - 1.5 dc), so where a root and a second sign
 * change of the secular function lie closer together than a scan step, whether the reference sees a sign change at all
 * depends on that root to the last bit -- its own outcome flips under a 1e-6 perturbation, and the short sequence's roots
 * differ from the reference's by up to that.  The one mechanism found (9.4 million random models): a root
 * creeping up to a half-space velocity has a mirror image just above it (the half-space terms use |k - k_beta|).  The
 * guard: a model whose short-sequence run (a) fails in any mode or (b) accepts a root within two scan steps of the fastest
 * S velocity or of a half-space velocity is run AGAIN with the reference's sequence, and that result is the one returned:
 * failure flags and the zero-from-period-k rows are then the reference's by construction, velocities its bits. */
int bho_surfdisp96(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                   int nlayer, int iflsph, int iwave, int mode, int igr, int kmax,
                   const double *t, double *cg, int64_t *neval)
{
    const int fm = g_fast_search;
    int guard = 0;
    g_neval = 0;
    int err = surfdisp96_run(thkm, vpm, vsm, rhom, nlayer, iflsph, iwave, mode, igr, kmax, t, cg, fm != 0,
                             fm == 2 ? &guard : NULL);
    if (guard) {
#pragma omp atomic
        g_guarded += 1;
        err = surfdisp96_run(thkm, vpm, vsm, rhom, nlayer, iflsph, iwave, mode, igr, kmax, t, cg, 0, NULL);
    }
    if (neval) *neval = g_neval;
    return err;
}

void bho_swd_batch(int B, int Lmax, const int32_t *nlay, const double *h, const double *vp,
                   const double *vs, const double *rho, int K, const double *periods,
                   int iwave, int igr, int mode, int flsph, double *vel, int32_t *err,
                   int64_t *neval_total, int nthreads)
{
    int64_t total = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : total)
    for (int ib = 0; ib < B; ++ib) {
        float fh[NLMAX], fvp[NLMAX], fvs[NLMAX], frho[NLMAX];
        double cg[NPMAX];
        int n = nlay[ib];
        for (int i = 0; i < n; ++i) {
            fh[i] = (float)h[(size_t)ib * Lmax + i];
            fvp[i] = (float)vp[(size_t)ib * Lmax + i];
            fvs[i] = (float)vs[(size_t)ib * Lmax + i];
            frho[i] = (float)rho[(size_t)ib * Lmax + i];
        }
        int64_t ne = 0;
        int e = bho_surfdisp96(fh, fvp, fvs, frho, n, flsph, iwave, mode, igr, K, periods, cg, &ne);
        total += ne;
        err[ib] = e;
        for (int k = 0; k < K; ++k) vel[(size_t)ib * K + k] = cg[k];
    }
    if (neval_total) *neval_total = total;
}
