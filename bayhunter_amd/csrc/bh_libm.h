/* Bit-exact restatement of glibc 2.35's x86-64 `sincos` (dbl-64/s_sincos.c + s_sin.c, the
 * IBM Accurate Mathematical Library path; the symbol is not multiarch: no FMA) and of `exp`
 * (dbl-64/e_exp.c, the FMA ifunc variant that x86-64 CPUs with FMA+AVX2 select), as found in
 * this image's /lib/x86_64-linux-gnu/libm.so.6.  Constants and tables were read out of that
 * binary (tools/libm_port/extract.py); the operation order follows the disassembly.
 * The same text compiles for the host (validation against libm) and for the device.
 * Compile with -ffp-contract=off: every FMA below is explicit. */
#ifndef BH_LIBM_PORT_H
#define BH_LIBM_PORT_H
#include <stdint.h>
#ifndef BH_HD
#define BH_HD static inline
#endif
#ifndef BH_TAB
#define BH_TAB static const
#endif
#include "bh_libm_tables.inc"

BH_HD double bhp_asdouble(uint64_t u) { union { uint64_t u; double d; } c; c.u = u; return c.d; }
BH_HD uint64_t bhp_asuint(double d) { union { uint64_t u; double d; } c; c.d = d; return c.u; }

/* ---- exp: e_exp.c (FMA contraction pattern of __exp_fma) ------------------------------------ */
BH_HD int bhp_exp_in_domain(double x) /* 2^-54 <= |x| < 512: the table path without special cases */
{
    const uint32_t abstop = (uint32_t)(bhp_asuint(x) >> 52) & 0x7ff;
    return (abstop - 0x3c9u) <= 0x3eu;
}
BH_HD double bhp_exp_core(double x, const uint64_t *T)
{
    const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52;
    const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5,
                 C5 = 0x1.1111167a4d017p-7;
    double kd = __builtin_fma(x, InvLn2N, Shift);
    const uint64_t ki = bhp_asuint(kd);
    kd = kd - Shift;
    double r = __builtin_fma(kd, NegLn2hiN, x);
    r = __builtin_fma(kd, NegLn2loN, r);
    const uint64_t idx = 2 * (ki & 127);
    const uint64_t top = ki << 45;
    const double tail = bhp_asdouble(T[idx]);
    const uint64_t sbits = T[idx + 1] + top;
    const double r2 = r * r;
    const double p23 = __builtin_fma(C3, r, C2);
    const double rt = r + tail;
    const double p45 = __builtin_fma(r, C5, C4);
    const double acc = __builtin_fma(p23, r2, rt);
    const double r4 = r2 * r2;
    const double tmp = __builtin_fma(r4, p45, acc);
    const double scale = bhp_asdouble(sbits);
    return __builtin_fma(scale, tmp, scale);
}

/* ---- sincos: s_sin.c / s_sincos.c ------------------------------------------------------------ */
#define BHP_BIG 0x1.8p45
#define BHP_SN3 (-0x1.5555555555515p-3)
#define BHP_SN5 0x1.11110e829872fp-7
#define BHP_CS2 0.5
#define BHP_CS4 (-0x1.5555555555535p-5)
#define BHP_CS6 0x1.6c16bedd9e239p-10

BH_HD double bhp_taylor_sin(double xx, double a, double da)
{
    const double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13,
                 s4 = 0x1.71de27b9a7ed9p-19, s5 = -0x1.addffc2fcdf59p-26;
    const double p2 = (((s5 * xx + s4) * xx + s3) * xx + s2) * xx;
    const double t = ((p2 + s1) * a - 0.5 * da) * xx + da;
    return a + t;
}
BH_HD double bhp_do_cos(double x, double dx, const double *tab)
{
    if (x < 0) dx = -dx;
    const double ax = __builtin_fabs(x);
    const double u = BHP_BIG + ax;
    x = ax - (u - BHP_BIG) + dx;
    const double xx = x * x;
    const double s = x + x * xx * (BHP_SN3 + xx * BHP_SN5);
    const double c = xx * (BHP_CS2 + xx * (BHP_CS4 + xx * BHP_CS6));
    const int k = (int)(uint32_t)bhp_asuint(u) * 4;
    const double sn = tab[k], ssn = tab[k + 1], cs = tab[k + 2], ccs = tab[k + 3];
    const double cor = (ccs - s * ssn - cs * c) - sn * s;
    return cs + cor;
}
BH_HD double bhp_do_sin(double x, double dx, const double *tab)
{
    const double xold = x;
    if (__builtin_fabs(x) < 0.126) return bhp_taylor_sin(x * x, x, dx);
    if (x <= 0) dx = -dx;
    const double ax = __builtin_fabs(x);
    const double u = BHP_BIG + ax;
    x = ax - (u - BHP_BIG);
    const double xx = x * x;
    const double s = x + (dx + x * xx * (BHP_SN3 + xx * BHP_SN5));
    const double c = x * dx + xx * (BHP_CS2 + xx * (BHP_CS4 + xx * BHP_CS6));
    const int k = (int)(uint32_t)bhp_asuint(u) * 4;
    const double sn = tab[k], ssn = tab[k + 1], cs = tab[k + 2], ccs = tab[k + 3];
    const double cor = (ssn + s * ccs - sn * c) + cs * s;
    return __builtin_copysign(sn + cor, xold);
}
BH_HD double bhp_do_sincos(double a, double da, int n, const double *tab)
{
    const double r = (n & 1) ? bhp_do_cos(a, da, tab) : bhp_do_sin(a, da, tab);
    return (n & 2) ? -r : r;
}
/* returns 0 when |x| is outside the restated range (>= 105414350, inf, nan): caller falls back */
BH_HD int bhp_sincos(double x, double *sn, double *cs, const double *tab)
{
    const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
    const int k = (int)((bhp_asuint(x) >> 32) & 0x7fffffff);
    if (k < 0x400368fd) {
        if (k < 0x3e400000) { /* |x| < 2^-27 */
            *sn = x;
            *cs = 1.0;
            return 1;
        }
        if (k < 0x3feb6000) { /* |x| < 0.855469 */
            *sn = bhp_do_sin(x, 0.0, tab);
            *cs = bhp_do_cos(x, 0.0, tab);
            return 1;
        }
        const double y = hp0 - __builtin_fabs(x); /* |x| < 2.426265 */
        const double a = y + hp1;
        const double da = (y - a) + hp1;
        *sn = __builtin_copysign(bhp_do_cos(a, da, tab), x);
        *cs = bhp_do_sin(a, da, tab);
        return 1;
    }
    if (k < 0x419921FB) { /* |x| < 105414350: reduce_sincos */
        const double mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27,
                     pp3 = -0x1.cb3b398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
        const double hpinv = 0x1.45f306dc9c883p-1, toint = 0x1.8p52;
        const double t = x * hpinv + toint;
        const double xn = t - toint;
        const double y = (x - xn * mp1) - xn * mp2;
        const int n = (int)(uint32_t)bhp_asuint(t) & 3;
        double t1 = xn * pp3;
        const double t2 = y - t1;
        double db = (y - t2) - t1;
        t1 = xn * pp4;
        const double b = t2 - t1;
        db += (t2 - b) - t1;
        *sn = bhp_do_sincos(b, db, n, tab);
        *cs = bhp_do_sincos(b, db, n + 1, tab);
        return 1;
    }
    return 0;
}

/* Branch-light form of bhp_sincos for SIMT execution: the three argument ranges differ only in
 * how the reduced argument (A, DA) and the quadrant are obtained, so all three are formed and
 * selected, then ONE do_sin and ONE do_cos evaluation (sharing the table entry) serve both
 * outputs.  Every selected value is produced by exactly the operations of the branchy form above,
 * so the results are the same bits (tools/libm_port/test_port.c checks both against libm).
 * |x| < 2^-27 needs no special case: the general path rounds to sin = x, cos = 1 there. */
BH_HD int bhp_sincos_bl(double x, double *sn_out, double *cs_out, const double *tab)
{
    const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
    const double mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27,
                 pp3 = -0x1.cb3b398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
    const double hpinv = 0x1.45f306dc9c883p-1, toint = 0x1.8p52;
    const int k = (int)((bhp_asuint(x) >> 32) & 0x7fffffff);
    /* range 3: reduce_sincos */
    const double t = x * hpinv + toint;
    const double xn = t - toint;
    const double y = (x - xn * mp1) - xn * mp2;
    const int n3 = (int)(uint32_t)bhp_asuint(t) & 3;
    double t1 = xn * pp3;
    const double t2 = y - t1;
    double db = (y - t2) - t1;
    t1 = xn * pp4;
    const double b = t2 - t1;
    db += (t2 - b) - t1;
    /* range 2: pi/2 - |x| */
    const double y2 = hp0 - __builtin_fabs(x);
    const double a2 = y2 + hp1;
    const double da2 = (y2 - a2) + hp1;
    const int r1 = k < 0x3feb6000, r2 = !r1 && k < 0x400368fd;
    const double A = r1 ? x : (r2 ? a2 : b);
    const double DA = r1 ? 0.0 : (r2 ? da2 : db);
    /* shared by do_sin and do_cos */
    const double ax = __builtin_fabs(A);
    const double u = BHP_BIG + ax;
    const double xr = ax - (u - BHP_BIG);
    const int ti = (int)(uint32_t)bhp_asuint(u) * 4;
    const double sn = tab[ti], ssn = tab[ti + 1], cs = tab[ti + 2], ccs = tab[ti + 3];
    /* do_sin(A, DA) */
    double S;
    {
        const double dxs = (A <= 0) ? -DA : DA;
        const double xx = xr * xr;
        const double s = xr + (dxs + xr * xx * (BHP_SN3 + xx * BHP_SN5));
        const double c = xr * dxs + xx * (BHP_CS2 + xx * (BHP_CS4 + xx * BHP_CS6));
        const double cor = (ssn + s * ccs - sn * c) + cs * s;
        const double tabres = __builtin_copysign(sn + cor, A);
        const double tay = bhp_taylor_sin(A * A, A, DA);
        S = (ax < 0.126) ? tay : tabres;
    }
    /* do_cos(A, DA) */
    double Cc;
    {
        const double dxc = (A < 0) ? -DA : DA;
        const double xc = xr + dxc;
        const double xx = xc * xc;
        const double s = xc + xc * xx * (BHP_SN3 + xx * BHP_SN5);
        const double c = xx * (BHP_CS2 + xx * (BHP_CS4 + xx * BHP_CS6));
        const double cor = (ccs - s * ssn - cs * c) - sn * s;
        Cc = cs + cor;
    }
    /* assemble */
    const int n = (r1 || r2) ? 0 : n3;
    double so = (n & 1) ? Cc : S;
    double co = (n & 1) ? S : Cc;
    so = (n & 2) ? -so : so;
    co = ((n + 1) & 2) ? -co : co;
    if (r2) {
        so = __builtin_copysign(Cc, x);
        co = S;
    }
    *sn_out = so;
    *cs_out = co;
    return k < 0x419921FB;
}

/* ---- log: dbl-64/e_log.c, the FMA ifunc variant (__log_fma) ------------------------------------
 * bhp_log_data: ln2hi, ln2lo, A[5] (poly), B[11] (poly1, B[0] = -0.5), then tab[128] of (invc, logc).
 * Returns 0 for arguments the table paths do not cover (x <= 0, subnormal, inf, nan). */
BH_HD int bhp_log(double x, double *out, const uint64_t *D)
{
    const double ln2hi = bhp_asdouble(D[0]), ln2lo = bhp_asdouble(D[1]);
    const uint64_t *A = D + 2, *B = D + 7, *T = D + 18;
    const uint64_t ix = bhp_asuint(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    if (ix - 0x3fee000000000000ull <= 0x308ffffffffffull) { /* 1 - 2^-4 <= x < 1 + 0x1.09p-4 */
        if (ix == 0x3ff0000000000000ull) {
            *out = 0.0;
            return 1;
        }
        const double r = x - 1.0;
        const double p1 = __builtin_fma(r, bhp_asdouble(B[2]), bhp_asdouble(B[1]));
        const double p4 = __builtin_fma(r, bhp_asdouble(B[5]), bhp_asdouble(B[4]));
        const double r2 = r * r;
        const double p7 = __builtin_fma(r, bhp_asdouble(B[8]), bhp_asdouble(B[7]));
        const double q1 = __builtin_fma(r2, bhp_asdouble(B[3]), p1);
        const double q4 = __builtin_fma(r2, bhp_asdouble(B[6]), p4);
        const double r3 = r * r2;
        double q7 = __builtin_fma(r2, bhp_asdouble(B[9]), p7);
        q7 = __builtin_fma(r3, bhp_asdouble(B[10]), q7);
        double poly = __builtin_fma(q7, r3, q4);
        poly = __builtin_fma(poly, r3, q1);
        const double two27 = 0x1p27;
        const double w = __builtin_fma(r, two27, r);
        const double rhi = __builtin_fma(-two27, r, w);
        const double b0 = bhp_asdouble(B[0]);
        const double rhi2 = rhi * rhi;
        const double rlo = r - rhi;
        const double hi = __builtin_fma(rhi2, b0, r);
        const double rmhi = r - hi;
        const double rsum = r + rhi;
        double lo = __builtin_fma(rhi2, b0, rmhi);
        const double b0rlo = b0 * rlo;
        lo = __builtin_fma(b0rlo, rsum, lo);
        const double y = __builtin_fma(poly, r3, lo);
        *out = y + hi;
        return 1;
    }
    if (top - 0x0010u > 0x7fdfu) return 0; /* zero, subnormal, negative, inf, nan */
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int64_t k = (int64_t)tmp >> 52;
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    const double invc = bhp_asdouble(T[2 * i]), logc = bhp_asdouble(T[2 * i + 1]);
    const double z = bhp_asdouble(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double kd = (double)(int32_t)k;
    const double w = __builtin_fma(kd, ln2hi, logc);
    const double p = __builtin_fma(r, bhp_asdouble(A[2]), bhp_asdouble(A[1]));
    const double hi = w + r;
    const double r2 = r * r;
    double lo = w - hi;
    lo = lo + r;
    lo = __builtin_fma(kd, ln2lo, lo);
    const double r3 = r * r2;
    double q = __builtin_fma(r, bhp_asdouble(A[4]), bhp_asdouble(A[3]));
    lo = __builtin_fma(r2, bhp_asdouble(A[0]), lo);
    q = __builtin_fma(q, r2, p);
    const double y = __builtin_fma(r3, q, lo);
    *out = y + hi;
    return 1;
}

/* ---- powf: flt-32/e_powf.c, the FMA ifunc variant (__powf_fma), main path only ----------------------
 * x positive and normal, y finite and non-zero, |y*log2(x)| < 126.  bhp_powf_log2_data: tab[16] of
 * (invc, logc), A[5]; bhp_exp2f_data: tab[32], shift_scaled, C[3].  Returns 0 outside that domain. */
BH_HD int bhp_powf(float x, float y, float *out, const uint64_t *L, const uint64_t *E)
{
    union { float f; uint32_t u; } cx, cy, cz;
    cx.f = x;
    cy.f = y;
    const uint32_t ix = cx.u, iy = cy.u;
    if (ix - 0x00800000u > 0x7effffffu) return 0;       /* x < 2^-126, negative, inf or nan */
    if (2u * iy - 1u > 0xfefffffeu) return 0;            /* y zero, inf or nan */
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15);
    const uint32_t topb = tmp & 0xff800000u;
    cz.u = ix - topb;
    const int32_t k = (int32_t)topb >> 23;
    const double invc = bhp_asdouble(L[2 * i]), logc = bhp_asdouble(L[2 * i + 1]);
    const uint64_t *A = L + 32;
    const double z = (double)cz.f;
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double ya = __builtin_fma(r, bhp_asdouble(A[0]), bhp_asdouble(A[1]));
    const double p = __builtin_fma(r, bhp_asdouble(A[2]), bhp_asdouble(A[3]));
    const double r2 = r * r;
    double q = __builtin_fma(r, bhp_asdouble(A[4]), y0);
    const double r4 = r2 * r2;
    q = __builtin_fma(r2, p, q);
    const double logx = __builtin_fma(ya, r4, q);
    const double ylogx = (double)y * logx;
    if (((bhp_asuint(ylogx) >> 47) & 0xffff) > 0x80beu) return 0; /* |ylogx| >= 126: over/underflow handling */
    const double shift = bhp_asdouble(E[32]);
    const uint64_t *C = E + 33;
    double kd = ylogx + shift;
    const uint64_t ki = bhp_asuint(kd);
    kd = kd - shift;
    const double rr = ylogx - kd;
    const uint64_t t = E[ki & 31] + (ki << 47);
    const double zz = __builtin_fma(rr, bhp_asdouble(C[0]), bhp_asdouble(C[1]));
    const double rr2 = rr * rr;
    double yy = __builtin_fma(rr, bhp_asdouble(C[2]), 1.0);
    yy = __builtin_fma(zz, rr2, yy);
    yy = yy * bhp_asdouble(t);
    *out = (float)yy;
    return 1;
}

#endif
