// bayhunter_amd/csrc/swd_fa.h -- the FAST ARITHMETIC of the dispersion kernels (included by swd_common.h; FA builds of
// swd_group_kernel / swd_kernel; bh_engine_set_swd_arith).
//
// The same secular functions as the reference-exact ones of swd_common.h (surfdisp96.f:710-1068: dltar1, dltar4, var, dnka,
// normc) -- the same formulas, the same per-layer max-norm scaling -- evaluated with what the chip does quickly instead of
// with the reference's rounding points: fused multiply-adds, reciprocals and square roots from the hardware seed
// (v_rcp_f64 / v_rsq_f64) refined by Newton / Goldschmidt steps, sin / cos / exp from two-part Cody-Waite reductions and
// polynomials, e^-(p+q) as a product of the two wave types' exponentials.  Every result is within a few
// units in the last place of the exactly rounded formula, i.e. it differs from the reference-exact evaluation by what that
// evaluation's own rounding error is; about a third of its instructions.
//
// What that does to a search (SearchT, short refinement; swd_lean.hip).  Measured on the device over 10^8 (model, period,
// velocity) points of each wave type (tools/ubench/fadiff.hip), both values max-norm scaled (|f| <= 1): |f_fast - f_exact| is
// below 1e-11 in 99.6 % of them and reaches 1.2e-8 -- the large ones where the trial velocity lies within ~1e-7 relative of a
// layer velocity, where the REFERENCE's own k - k_beta cancels and its value is uncertain by as much.  A root moves by ~1e-13
// relative (tolerance of the path: 1e-5; the short refinement's own distance from the reference: 1.2e-6).  A scan's sign
// pattern can differ from the exact evaluation's only where |f| at a grid point is smaller than that difference, i.e. where
// a root lies within ~1e-8 of the grid point: the bracket then moves by one step around the same root (same velocity, same
// flag).  What the kernels do NOT accept is a value that is not a number or is exactly lost (|f| < SIGN_FLOOR = 1e-12: an
// argument beyond the sin / cos reduction's range, an underflow): it fires the short refinement's guard and the model is run
// again with the reference's sequence in the reference's arithmetic -- the path a guarded model takes anyway.  The parity
// statement rests on the tests: failure flags and zero rows identical to the reference's on 1.7 million LVZ-rich models
// (tests/test_gpu_swd_lean.py), velocities within 2e-6.
#pragma once

namespace fa {
constexpr double SIGN_FLOOR = 1.0e-12;

// sqrt(x) and 1/sqrt(x) together (x > 0, normal): hardware seed (~2^-23), two Goldschmidt steps
__device__ __forceinline__ void sqrt_rsqrt(double x, double &s, double &rs)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    s = g;
    rs = h + h;
}
__device__ __forceinline__ double rcp(double x) // hardware seed, two Newton steps
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
__device__ __forceinline__ double rcp1(double x) // one step (~2^-45): scale factors whose error cancels or does not matter
{
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}

// sin and cos of 0 <= x < 1e5 and exp(-x), 0 <= x <= 700: fdlibm's kernels on
// [-pi/4, pi/4]; Taylor to r^12 on |r| <= ln 2 / 2), the polynomials summed by Estrin's scheme -- half the depth of the chain of
// dependent operations, which is what two wavefronts per SIMD cannot hide.
__device__ __forceinline__ void sincos(double x, double &sn, double &cs)
{
    const double n = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-n, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-n, 6.07710050650619224932e-11, r);
    const double z = r * r, z2 = z * z, z4 = z2 * z2;
    const double s01 = __builtin_fma(z, 8.33333333332248946124e-03, -1.66666666666666324348e-01);
    const double s23 = __builtin_fma(z, 2.75573137070700676789e-06, -1.98412698298579493134e-04);
    const double s45 = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    const double ps = __builtin_fma(z4, s45, __builtin_fma(z2, s23, s01));
    const double s = __builtin_fma(r * z, ps, r);
    const double c01 = __builtin_fma(z, -1.38888888888741095749e-03, 4.16666666666666019037e-02);
    const double c23 = __builtin_fma(z, -2.75573143513906633035e-07, 2.48015872894767294178e-05);
    const double c45 = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    const double pc = __builtin_fma(z4, c45, __builtin_fma(z2, c23, c01));
    const double c = __builtin_fma(z2, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)n & 3;
    const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    sn = (q & 2) ? -ss : ss;
    cs = ((q + 1) & 2) ? -cc : cc;
}
__device__ __forceinline__ double expneg(double x)
{
    const double t = -x;
    const double n = __builtin_rint(t * 1.44269504088896338700e+00);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, t);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    // p(r) = 1/2 + r/6 + r^2/24 + ... + r^10/12! (so that e^r = 1 + r + r^2 p)
    const double p01 = __builtin_fma(r, 1.66666666666666666667e-01, 0.5);
    const double p23 = __builtin_fma(r, 8.33333333333333333333e-03, 4.16666666666666666667e-02);
    const double p45 = __builtin_fma(r, 1.98412698412698412698e-04, 1.38888888888888888889e-03);
    const double p67 = __builtin_fma(r, 2.75573192239858906526e-06, 2.48015873015873015873e-05);
    const double p89 = __builtin_fma(r, 2.50521083854417187751e-08, 2.75573192239858906526e-07);
    const double p03 = __builtin_fma(r2, p23, p01), p47 = __builtin_fma(r2, p67, p45);
    const double p8a = __builtin_fma(r2, 2.08767569878680989792e-09, p89);
    const double p = __builtin_fma(r8, p8a, __builtin_fma(r4, p47, p03));
    const double e = __builtin_fma(r2, p, r) + 1.0;
    return e * __longlong_as_double((long long)((unsigned long long)((long long)n + 1023ll) << 52));
}

// One wave type of one layer (surfdisp96.f:906-935): cos-like, sin-like / r, -+ r sin-like, and e^-p of an evanescent wave
// (1 for a propagating one).  k == xk takes the reference's limits (cos = 1, w = d, x = 0).  Arguments beyond the
// reduction's range poison the value (NaN -> the guard).
__device__ __forceinline__ void wave(double k, double xk, double d, double &cs, double &w, double &x, double &em)
{
    const double dk = k - xk;
    const double r2 = (k + xk) * fabs(dk);
    const bool zero = !(r2 > 1.0e-290);
    double r, rr;
    sqrt_rsqrt(zero ? 1.0 : r2, r, rr);
    r = zero ? 0.0 : r;
    const double p = r * d;
    double sn;
    if (dk < 0.0) {
        sincos(fmin(p, 9.0e4), sn, cs);
        x = -(r * sn);
        em = 1.0;
    } else {
        em = expneg(fmin(p, 700.0));
        const double hf = 0.5 * (em * em);
        cs = 0.5 + hf;
        sn = 0.5 - hf;
        x = r * sn;
    }
    w = zero ? d : sn * rr;
    if (!(p < 9.0e4)) cs = __builtin_nan("");
}

// var (surfdisp96.f:874-991): the eigenfunction products of one layer
__device__ __forceinline__ void layer_products(double wvno, double xka, double xkb, double dpth, LayerTerms &o)
{
    double cosp, w, x, ep, cosq, y, z, eq;
    wave(wvno, xka, dpth, cosp, w, x, ep);
    wave(wvno, xkb, dpth, cosq, y, z, eq);
    o.a0 = ep * eq;
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

// dnka (surfdisp96.f:1024-1068), irho = 1 / rho
__device__ __forceinline__ void ca19(Ca19 &o, double wvno2, double gam, double gammk, double rho, double irho, const LayerTerms &v)
{
#pragma clang fp contract(fast)
    const double gamm1 = gam - 1.0;
    const double twgm1 = gam + gamm1;
    const double gmgmk = gam * gammk;
    const double gmgm1 = gam * gamm1;
    const double gm1sq = gamm1 * gamm1;
    const double rho2 = rho * rho, irho2 = irho * irho;
    const double a0pq = v.a0 - v.cpcq;
    const double ca11 = v.cpcq - 2.0 * gmgm1 * a0pq - gmgmk * v.xz - wvno2 * gm1sq * v.wy;
    const double ca12 = (wvno2 * v.cpy - v.cqx) * irho;
    const double ca13 = -(twgm1 * a0pq + gammk * v.xz + wvno2 * gamm1 * v.wy) * irho;
    const double ca14 = (v.cpz - wvno2 * v.cqw) * irho;
    const double ca15 = -(2.0 * wvno2 * a0pq + v.xz + wvno2 * wvno2 * v.wy) * irho2;
    const double ca21 = (gmgmk * v.cpz - gm1sq * v.cqw) * rho;
    const double ca23 = gammk * v.cpz - gamm1 * v.cqw;
    const double ca41 = (gm1sq * v.cpy - gmgmk * v.cqx) * rho;
    const double ca43 = gamm1 * v.cpy - gammk * v.cqx;
    const double ca51 = -(2.0 * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * v.xz + gm1sq * gm1sq * v.wy) * rho2;
    const double ca53 = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * v.xz + gamm1 * gm1sq * v.wy) * rho;
    const double t = -2.0 * wvno2;
    o.c[0] = ca11; o.c[1] = ca12; o.c[2] = ca13; o.c[3] = ca14; o.c[4] = ca15;
    o.c[5] = ca21; o.c[6] = ca23; o.c[7] = -v.wz; o.c[8] = v.cpcq;
    o.c[9] = ca41; o.c[10] = -v.xy; o.c[11] = ca43;
    o.c[12] = ca51; o.c[13] = ca53;
    o.c[14] = t * ca53;
    o.c[15] = t * ca43;
    o.c[16] = v.a0 + 2.0 * (v.cpcq - ca11);
    o.c[17] = t * ca23;
    o.c[18] = t * ca13;
}

// the half-space vector (surfdisp96.f:800-808)
__device__ __forceinline__ void rayleigh_halfspace(double e[5], double wvno, double wvno2, double xka, double xkb, double gammk, double rho1)
{
#pragma clang fp contract(fast)
    const double ra2 = (wvno + xka) * fabs(wvno - xka), rb2 = (wvno + xkb) * fabs(wvno - xkb);
    double ra, rb, t;
    sqrt_rsqrt(ra2 > 1.0e-290 ? ra2 : 1.0, ra, t);
    sqrt_rsqrt(rb2 > 1.0e-290 ? rb2 : 1.0, rb, t);
    ra = ra2 > 1.0e-290 ? ra : 0.0;
    rb = rb2 > 1.0e-290 ? rb : 0.0;
    const double gam = gammk * wvno2;
    const double gamm1 = gam - 1.0;
    const double rarb = ra * rb;
    e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * rarb);
    e[1] = -rho1 * ra;
    e[2] = rho1 * (gamm1 - gammk * rarb);
    e[3] = rho1 * rb;
    e[4] = wvno2 - rarb;
}

// normc (surfdisp96.f:995-1020)
__device__ __forceinline__ void normalize5(double v0, double v1, double v2, double v3, double v4, double e[5])
{
    double t1 = fmax(fmax(fmax(fabs(v0), fabs(v1)), fmax(fabs(v2), fabs(v3))), fabs(v4));
    if (t1 < 1.0e-40) t1 = 1.0;
    const double r = rcp1(t1);
    e[0] = v0 * r;
    e[1] = v1 * r;
    e[2] = v2 * r;
    e[3] = v3 * r;
    e[4] = v4 * r;
}

// one layer of the Love recursion (surfdisp96.f:758-767) on yx = y / xmu, zx = xmu * z (formed with the layer terms)
__device__ __forceinline__ void love_step(double &e1, double &e2, double cosq, double yx, double zx)
{
    const double e10 = __builtin_fma(e2, zx, e1 * cosq);
    const double e20 = __builtin_fma(e1, yx, e2 * cosq);
    double xnor = fmax(fabs(e10), fabs(e20));
    if (xnor < 1.0e-40) xnor = 1.0;
    const double r = rcp1(xnor);
    e1 = e10 * r;
    e2 = e20 * r;
}
// the terms of one Love layer: cosq, y, z (surfdisp96.f:738-757)
__device__ __forceinline__ void love_terms(double wvno, double xkb, double dm, double &cosq, double &y, double &z, double &q)
{
    double em;
    const double dk = wvno - xkb;
    const double r2 = (wvno + xkb) * fabs(dk);
    const bool zero = !(r2 > 1.0e-290);
    double rb, rr;
    sqrt_rsqrt(zero ? 1.0 : r2, rb, rr);
    rb = zero ? 0.0 : rb;
    q = dm * rb;
    double sn;
    if (dk < 0.0) {
        sincos(fmin(q, 9.0e4), sn, cosq);
        z = -(rb * sn);
    } else {
        em = expneg(fmin(q, 700.0));
        const double hf = 0.5 * (em * em);
        cosq = 0.5 + hf;
        sn = 0.5 - hf;
        z = rb * sn;
    }
    y = zero ? dm : sn * rr;
    if (!(q < 9.0e4)) cosq = __builtin_nan("");
}

// ---- the whole secular functions, all layers serial in one lane (swd_kernel.hip) ---------------------------------------
// e <- e * CA (surfdisp96.f:836-842), the 19 distinct entries as rayleigh_ca19 stores them
__device__ __forceinline__ void apply5(const double e[5], const double *c, double ee[5])
{
    const double ca11 = c[0], ca12 = c[1], ca13 = c[2], ca14 = c[3], ca15 = c[4];
    const double ca21 = c[5], ca23 = c[6], ca24 = c[7], ca22 = c[8];
    const double ca41 = c[9], ca42 = c[10], ca43 = c[11], ca51 = c[12], ca53 = c[13];
    const double ca31 = c[14], ca32 = c[15], ca33 = c[16], ca34 = c[17], ca35 = c[18];
    ee[0] = __builtin_fma(e[4], ca51, __builtin_fma(e[2], ca31, e[0] * ca11)) + __builtin_fma(e[3], ca41, e[1] * ca21);
    ee[1] = __builtin_fma(e[4], ca41, __builtin_fma(e[2], ca32, e[0] * ca12)) + __builtin_fma(e[3], ca42, e[1] * ca22);
    ee[2] = __builtin_fma(e[4], ca53, __builtin_fma(e[2], ca33, e[0] * ca13)) + __builtin_fma(e[3], ca43, e[1] * ca23);
    ee[3] = __builtin_fma(e[4], ca21, __builtin_fma(e[2], ca34, e[0] * ca14)) + __builtin_fma(e[3], ca22, e[1] * ca24);
    ee[4] = __builtin_fma(e[4], ca11, __builtin_fma(e[2], ca35, e[0] * ca15)) + __builtin_fma(e[3], ca12, e[1] * ca14);
}

// dltar4 (surfdisp96.f:773-871)
template <class MD>
__device__ __forceinline__ double rayleigh_secular(double wvno, double omga, const MD &md, int mmax, int llw, int mtop)
{
    const double omega = omga < 1.0e-4 ? 1.0e-4 : omga;
    const double wvno2 = wvno * wvno;
    const double iom = rcp(omega);
    double e[5];
    {
        const double bh = md.Bv(mmax - 1);
        const double t = bh * iom;
        rayleigh_halfspace(e, wvno, wvno2, omega * rcp(md.A(mmax - 1)), omega * rcp(bh), 2.0 * t * t, md.R(mmax - 1));
    }
    for (int m = mtop - 2; m >= 0; --m) {
        if (m <= mmax - 2 && m >= llw - 1) {
            const double am = md.A(m), bm = md.Bv(m), rho1 = md.R(m), dpth = md.D(m);
            const double t = bm * iom;
            const double gammk = 2.0 * t * t;
            LayerTerms v;
            layer_products(wvno, omega * rcp(am), omega * rcp(bm), dpth, v);
            Ca19 c;
            ca19(c, wvno2, gammk * wvno2, gammk, rho1, rcp(rho1), v);
            double ee[5];
            apply5(e, c.c, ee);
            normalize5(ee[0], ee[1], ee[2], ee[3], ee[4], e);
        }
    }
    double result = e[0];
    if (llw != 1) { // water layer on top (surfdisp96.f:850-866); unreachable from BayHunter
        LayerTerms v;
        layer_products(wvno, omega * rcp(md.A(0)), wvno + 1.0, md.D(0), v); // (only w and cosp of the P wave are used)
        result = v.cosp * e[0] - md.R(0) * v.w * e[1];
    }
    return result;
}

// dltar1 (surfdisp96.f:710-769); nv (optional): the packed mode count of this evaluation (LoveCount)
template <class MD>
__device__ __forceinline__ double love_secular(double wvno, double omega, const MD &md, int mmax, int llw, int mtop, int *nv = nullptr)
{
    double e1, e2;
    LoveCount lc;
    {
        const double ib = rcp(md.Bv(mmax - 1));
        const double xkb = omega * ib;
        const double r2 = (wvno + xkb) * fabs(wvno - xkb);
        double rb, t_;
        sqrt_rsqrt(r2 > 1.0e-290 ? r2 : 1.0, rb, t_);
        rb = r2 > 1.0e-290 ? rb : 0.0;
        e1 = md.R(mmax - 1) * rb;
        e2 = ib * ib;
        lc.reset(wvno > xkb);
    }
    for (int m = mtop - 2; m >= 0; --m) {
        if (m <= mmax - 2 && m >= llw - 1) {
            const double beta1 = md.Bv(m), rho1 = md.R(m);
            const double ib = rcp(beta1);
            const double xkb = omega * ib;
            double cosq, y, z, q;
            love_terms(wvno, xkb, md.D(m), cosq, y, z, q);
            const double fl = (wvno < xkb) ? love_zero_floor(q) : 0.0;
            const double e2o = e2;
            love_step(e1, e2, cosq, y * (ib * ib * rcp(rho1)), z * (rho1 * beta1 * beta1));
            lc.layer(fl, e2o, e2);
        }
    }
    if (nv != nullptr) *nv = lc.packed(e1, e2);
    return e1;
}
} // namespace fa
