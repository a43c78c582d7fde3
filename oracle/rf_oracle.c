/*
 * oracle/rf_oracle.c -- CPU restatement of the rfmini receiver-function path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity: pinned (tests/test_oracle_rf.py).
 *
 * Restates, in C99 complex arithmetic, what the reference's C++ does on the path
 *   rfmini/wrap.cpp:58-80 (synrf_cwrap) -> rfmini/synrf.cpp:16-55 (synrf)
 *   -> rfmini/model.cpp:221-252 (earth flattening, always on)
 *   -> rfmini/greens.cpp:400-591 (calcresp_core, non-derivative branch)
 *   -> rfmini/greens.cpp:343-398 (compute_rf) -> :136-158 (iftr) -> rfmini/fork.cpp:11-60.
 * The z/r traces, the SH response, bottom_up and every partial-derivative branch are
 * computed-but-discarded in the reference on this path and are not restated.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle.h"

typedef double complex cplx;

typedef struct { cplx c11, c12, c21, c22; } cmat2;          /* rfmini/cmat2.h:11-54 */

static inline cmat2 cm_mul(cmat2 x, cmat2 y)                 /* cmat2.h:171-178 */
{
    cmat2 r;
    r.c11 = x.c11 * y.c11 + x.c12 * y.c21;
    r.c12 = x.c11 * y.c12 + x.c12 * y.c22;
    r.c21 = x.c21 * y.c11 + x.c22 * y.c21;
    r.c22 = x.c21 * y.c12 + x.c22 * y.c22;
    return r;
}
static inline cmat2 cm_add(cmat2 x, cmat2 y)
{
    cmat2 r = {x.c11 + y.c11, x.c12 + y.c12, x.c21 + y.c21, x.c22 + y.c22};
    return r;
}
static inline cmat2 cm_sub(cmat2 x, cmat2 y)
{
    cmat2 r = {x.c11 - y.c11, x.c12 - y.c12, x.c21 - y.c21, x.c22 - y.c22};
    return r;
}
static inline cmat2 cm_inv(cmat2 x)                          /* cmat2.h:143-152 */
{
    cplx q = 1.0 / (x.c11 * x.c22 - x.c12 * x.c21);
    cmat2 r = {q * x.c22, -q * x.c12, -q * x.c21, q * x.c11};
    return r;
}
static inline cmat2 cm_scale_d(double s, cmat2 x)
{
    cmat2 r = {s * x.c11, s * x.c12, s * x.c21, s * x.c22};
    return r;
}
/* e*x*e for diagonal e, greens.cpp:829-845 */
static inline cmat2 cm_exe(cmat2 e, cmat2 x)
{
    cplx e11 = e.c11, e22 = e.c22;
    cplx e12 = e11 * e22;
    e11 = e11 * e11;
    e22 = e22 * e22;
    cmat2 r = {x.c11 * e11, x.c12 * e12, x.c21 * e12, x.c22 * e22};
    return r;
}

typedef struct { double z, h, vp, vs, rh, qp, qs; } flayer;

/* model.cpp:208-218 */
static int is_lower_halfspace(const flayer *l)
{
    if (l->h > 0.0) return 0;
    if (l->vp < 1.0 && l->rh < 0.1) return 0;
    return 1;
}

/* model.cpp:221-252, R = 6371 km */
static void flatten_layer(flayer *l)
{
    const double R = 6371.0;
    double zb = l->z + l->h;
    double r = R - l->z;
    double q = R / r;
    l->z = R * log(q);
    l->vp *= q;
    l->vs *= q;
    l->rh /= q;
    if (!is_lower_halfspace(l)) {
        r = R - zb;
        q = R / r;
        zb = R * log(q);
        l->h = zb - l->z;
    }
}

/* Solid-solid interface P/SV coefficients, greens.cpp:19-85 (SH part unused on this path). */
static void interface_coeffs(double u, double vp1, double vs1, double rho1, double vp2, double vs2,
                             double rho2, cmat2 *rd, cmat2 *td, cmat2 *ru, cmat2 *tu)
{
    double mue1 = rho1 * vs1 * vs1, mue2 = rho2 * vs2 * vs2;
    double c = 2. * (mue1 - mue2), u2 = u * u, cu2 = c * u2, t1, t2, t3;
    cplx rpp, rps, rsp, rss, tpp, tps, tsp, tss, d1, d2, t4, t5, t7;
    cplx a1 = conj(csqrt(CMPLX(1. / (vp1 * vp1) - u2, 0.0)));
    cplx a2 = conj(csqrt(CMPLX(1. / (vp2 * vp2) - u2, 0.0)));
    cplx b1 = conj(csqrt(CMPLX(1. / (vs1 * vs1) - u2, 0.0)));
    cplx b2 = conj(csqrt(CMPLX(1. / (vs2 * vs2) - u2, 0.0)));

    t1 = cu2 - rho1 + rho2;
    t2 = cu2 - rho1;
    t3 = cu2 + rho2;
    t4 = t3 * a1 - t2 * a2;

    /* incident from medium 1 (downward) */
    d1 = t1 * t1 * u2 + t2 * t2 * a2 * b2 + rho1 * rho2 * a2 * b1;
    d2 = c * c * u2 * a1 * a2 * b1 * b2 + t3 * t3 * a1 * b1 + rho1 * rho2 * a1 * b2;
    t5 = 1. / (d1 + d2);
    t7 = 2. * rho1 * t5;
    rpp = (d2 - d1) * t5;
    rps = -2. * u * a1 * t5 * (t1 * t3 + c * t2 * a2 * b2);
    tpp = a1 * t7 * (t3 * b1 - t2 * b2);
    tps = -a1 * t7 * u * (t1 + c * a2 * b1);
    rss = (d2 - d1 - 2. * rho1 * rho2 * (a1 * b2 - a2 * b1)) * t5;
    rsp = 2. * u * b1 * t5 * (t1 * t3 + c * t2 * a2 * b2);
    tss = b1 * t7 * t4;
    tsp = b1 * t7 * u * (t1 + c * a1 * b2);
    rd->c11 = rpp; rd->c12 = rsp; rd->c21 = rps; rd->c22 = rss;
    td->c11 = tpp; td->c12 = tsp; td->c21 = tps; td->c22 = tss;

    /* incident from medium 2 (upward) */
    d1 = t1 * t1 * u2 + t3 * t3 * a1 * b1 + rho1 * rho2 * a1 * b2;
    d2 = c * c * u2 * a1 * a2 * b1 * b2 + t2 * t2 * a2 * b2 + rho1 * rho2 * a2 * b1;
    t5 = 1. / (d1 + d2);
    t7 = 2. * rho2 * t5;
    rpp = (d2 - d1) * t5;
    rps = 2. * u * a2 * t5 * (t1 * t2 + c * t3 * a1 * b1);
    tpp = a2 * t7 * (t3 * b1 - t2 * b2);
    tps = -a2 * t7 * u * (t1 + c * a1 * b2);
    rss = (d2 - d1 - 2. * rho1 * rho2 * (a2 * b1 - a1 * b2)) * t5;
    rsp = -2. * u * b2 * t5 * (t1 * t2 + c * t3 * a1 * b1);
    tss = b2 * t7 * t4;
    tsp = b2 * t7 * u * (t1 + c * a2 * b1);
    ru->c11 = rpp; ru->c12 = rsp; ru->c21 = rps; ru->c22 = rss;
    tu->c11 = tpp; tu->c12 = tsp; tu->c21 = tps; tu->c22 = tss;
}

/* Free-surface reflection, greens.cpp:87-112 (note: plain sqrt, no conj) */
static void surface_coeffs(double u, double vp, double vs, cmat2 *ru)
{
    double u2 = u * u;
    cplx a = csqrt(CMPLX(1. / (vp * vp) - u2, 0.0));
    cplx b = csqrt(CMPLX(1. / (vs * vs) - u2, 0.0));
    cplx t1 = 2. * vs * vs;
    cplx t2 = t1 * u2 - 1.;
    cplx d1 = t2 * t2;
    cplx d2 = t1 * t1 * u2 * a * b;
    cplx d = d1 + d2;
    cplx t3 = 2. * t1 * u * t2 / d;
    cplx rpp = (d2 - d1) / d;
    ru->c11 = rpp;
    ru->c12 = -b * t3;
    ru->c21 = a * t3;
    ru->c22 = rpp;
}

/* greens.cpp:307-322 */
static void displacement_matrix(double p, double vp, double vs, cmat2 *m)
{
    double vp2 = vp * vp, vs2 = vs * vs, p2 = p * p, x = 1. - 2. * vs2 * p2;
    cplx a1 = conj(csqrt(CMPLX(1. / vp2 - p2, 0.0)));
    cplx b1 = conj(csqrt(CMPLX(1. / vs2 - p2, 0.0)));
    cplx q = 1. / (x * x + 4. * vs2 * vs2 * p2 * a1 * b1);
    m->c11 = q * a1 * b1 * 2. * vs2 * p;
    m->c12 = q * b1 * (1. - 2. * vs2 * p2);
    m->c21 = q * a1 * (1. - 2. * vs2 * p2);
    m->c22 = -q * a1 * b1 * 2. * vs2 * p;
}

/* rfmini/fork.cpp:11-60: radix-2 DIT, scaled by 1/sqrt(n) */
static void ccfork(int n, cplx *x, int signi)
{
    double sc = sqrt(1. / (double)n);
    int j = 0;
    for (int i = 0; i < n; ++i) {
        if (i <= j) {
            cplx tmp = x[j] * sc;
            x[j] = x[i] * sc;
            x[i] = tmp;
        }
        int m = n >> 1;
        do {
            if (j < m) break;
            j -= m;
            m >>= 1;
        } while (m >= 1);
        j += m;
    }
    int l = 1;
    do {
        int istep = 2 * l;
        for (int m = 0; m < l; ++m) {
            cplx w = cexp(CMPLX(0.0, M_PI * (double)(signi * m) / (double)l));
            for (int i = m; i < n; i += istep) {
                cplx tmp = w * x[i + l];
                x[i + l] = x[i] - tmp;
                x[i] += tmp;
            }
        }
        l = istep;
    } while (l < n);
}

int bho_synrf(int nsamp, double fsamp, double tshift, double p_in, double a, double nsv,
              double sigma, int waveno, int nlay, const double *z, const double *vp,
              const double *vs, const double *rh, const double *qp, const double *qs,
              double *rf)
{
    /* wrap.cpp:13,55,73-76 */
    const double vptop = nsv * sqrt((1. - sigma) / (.5 - sigma));
    const double vstop = nsv;
    const double p = p_in * 0.00899; /* s/deg -> s/km */
    const double fref = 1.0;         /* synrf.cpp:25 */
    const int nfreq = nsamp / 2 + 1;

    flayer *lay = (flayer *)malloc(sizeof(flayer) * (size_t)(nlay + 1));
    /* synrf.cpp:28-34; index 0 unused like in the reference */
    for (int i = 0; i < nlay - 1; ++i) {
        flayer l = {z[i], z[i + 1] - z[i], vp[i], vs[i], rh[i], qp[i], qs[i]};
        lay[i + 1] = l;
    }
    {
        flayer l = {z[nlay - 1], -1., vp[nlay - 1], vs[nlay - 1], rh[nlay - 1], qp[nlay - 1], qs[nlay - 1]};
        lay[nlay] = l;
    }
    for (int i = 1; i <= nlay; ++i) flatten_layer(&lay[i]);

    cmat2 *ru = (cmat2 *)calloc((size_t)(nlay + 2) * 8, sizeof(cmat2));
    cmat2 *rd = ru + (nlay + 2), *tu = rd + (nlay + 2), *td = tu + (nlay + 2);
    cmat2 *nb = td + (nlay + 2), *nt = nb + (nlay + 2), *g = nt + (nlay + 2), *e = g + (nlay + 2);
    cplx *cz = (cplx *)malloc(sizeof(cplx) * (size_t)nfreq * 3);
    cplx *cr = cz + nfreq, *crf = cr + nfreq;
    const double p2 = p * p;

    /* greens.cpp:462-468: coefficients from the real layer velocities, once per model */
    for (int i = 1; i <= nlay; ++i) {
        if (i == 1)
            surface_coeffs(p, lay[1].vp, lay[1].vs, &ru[1]); /* rd=td=tu=0 */
        else
            interface_coeffs(p, lay[i - 1].vp, lay[i - 1].vs, lay[i - 1].rh, lay[i].vp, lay[i].vs,
                             lay[i].rh, &rd[i], &td[i], &ru[i], &tu[i]);
    }
    cmat2 hm;
    displacement_matrix(p, lay[1].vp, lay[1].vs, &hm);

    const double wref = 2. * M_PI * fref;
    const double dw = 2.0 * M_PI * fsamp / nsamp;
    /* direct-wave travel time, greens.cpp:510-526 (half-space h = -1 included, as there) */
    double t0 = 0.;
    for (int i = 1; i <= nlay; ++i) {
        double v = (waveno == 0) ? lay[i].vp : lay[i].vs;
        t0 += lay[i].h * sqrt(1. / (v * v) - p2);
    }
    const cmat2 ident = {CMPLX(1.0, 0.0), CMPLX(0.0, 0.0), CMPLX(0.0, 0.0), CMPLX(1.0, 0.0)};

    for (int j = 0; j < nfreq; ++j) {
        double w = dw * j;
        double lgw = j ? log(w / wref) : 0;
        /* phase matrices, greens.cpp:533-549 */
        for (int i = 1; i <= nlay; ++i) {
            double d = lay[i].h;
            cplx miwd = CMPLX(0., -w * d);
            cplx vpc = lay[i].vp * (1. + lgw / (M_PI * lay[i].qp) + I / (2. * lay[i].qp));
            cplx vsc = lay[i].vs * (1. + lgw / (M_PI * lay[i].qs) + I / (2. * lay[i].qs));
            cplx plc = csqrt(1. / (vpc * vpc) - p2);
            cplx slc = csqrt(1. / (vsc * vsc) - p2);
            e[i].c11 = cexp(miwd * plc);
            e[i].c12 = 0;
            e[i].c21 = 0;
            e[i].c22 = cexp(miwd * slc);
        }
        /* top-down reflectivity recursion, greens.cpp:196-224 (options = 0) */
        cmat2 q = {0, 0, 0, 0};
        for (int i = 1; i < nlay; ++i) {
            if (i == 1)
                nt[i] = ru[1];
            else
                nt[i] = cm_add(ru[i], cm_mul(cm_mul(td[i], nb[i - 1]), q));
            nb[i] = cm_exe(e[i], nt[i]);
            q = cm_mul(cm_inv(cm_sub(ident, cm_mul(rd[i + 1], nb[i]))), tu[i + 1]);
            if (i == 1)
                g[i] = cm_mul(e[1], q);
            else
                g[i] = cm_mul(cm_mul(g[i - 1], e[i]), q);
        }
        cmat2 t = cm_mul(cm_scale_d(2.0, hm), g[nlay - 1]); /* t = 2*h*g[nlay-1], :572 */
        if (waveno == 0) { cr[j] = t.c11; cz[j] = t.c21; }
        else             { cr[j] = t.c12; cz[j] = t.c22; }
        cplx qq = cexp(CMPLX(0., w * t0));
        cr[j] *= qq;
        cz[j] *= qq;
    }

    /* compute_rf, greens.cpp:343-398 */
    {
        double qg = sqrt(M_PI) * fsamp / a;
        cplx *pz = cz, *pr = cr;
        if (vstop > 0.01 && fabs(p) > 0.0001) { /* decomp, :324-341, real a/b */
            double aa = sqrt(1. / (vptop * vptop) - p * p), bb = sqrt(1. / (vstop * vstop) - p * p);
            double m11 = -(2 * vstop * vstop * p * p - 1.) / (vptop * aa);
            double m12 = 2. * p * vstop * vstop / vptop;
            double m21 = -2. * p * vstop;
            double m22 = (1. - 2. * vstop * vstop * p * p) / (vstop * bb);
            for (int i = 0; i < nfreq; ++i) {
                cplx cx = cz[i] * m11 + cr[i] * m12;
                cplx cy = cz[i] * m21 + cr[i] * m22;
                cz[i] = cx;
                cr[i] = cy;
            }
        }
        if (waveno == 1) { cplx *tmp = pz; pz = pr; pr = tmp; } /* S-RF: deconvolve P with SV */
        for (int j = 0; j < nfreq; ++j) {
            double w = dw * j;
            double denom = creal(pz[j] * conj(pz[j]));
            cplx v = pr[j] * conj(pz[j]) / denom; /* no water level applied (:384 is commented out) */
            double wa = w / a;
            wa = (wa > 50.0) ? 50.0 : wa;
            cplx cq = qg * cexp(CMPLX(-0.25 * (wa * wa), -w * tshift));
            crf[j] = v * cq;
        }
    }
    /* iftr, greens.cpp:136-158 */
    {
        cplx *cx = (cplx *)malloc(sizeof(cplx) * (size_t)nsamp);
        double qn = 1. / sqrt((double)nsamp);
        for (int i = 0; i < nsamp / 2 + 1; ++i) cx[i] = crf[i];
        for (int i = nsamp / 2 + 1; i < nsamp; ++i) cx[i] = conj(cx[nsamp - i]);
        ccfork(nsamp, cx, 1);
        for (int i = 0; i < nsamp; ++i) rf[i] = qn * creal(cx[i]);
        free(cx);
    }
    free(cz);
    free(ru);
    free(lay);
    return 1;
}

void bho_rf_batch(int B, int Lmax, const int32_t *nlay, const double *h, const double *vp,
                  const double *vs, const double *rho, double p_s_per_deg, double gauss,
                  int nsamp, double fsamp, double tshift, int waveno, int nkeep,
                  double *rf_out, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        double *rf = (double *)malloc(sizeof(double) * (size_t)nsamp);
        double *zz = (double *)malloc(sizeof(double) * (size_t)Lmax * 3);
        double *qp = zz + Lmax, *qs = qp + Lmax;
#pragma omp for schedule(dynamic, 4)
        for (int ib = 0; ib < B; ++ib) {
            int n = nlay[ib];
            const double *hh = h + (size_t)ib * Lmax, *pvp = vp + (size_t)ib * Lmax;
            const double *pvs = vs + (size_t)ib * Lmax, *prh = rho + (size_t)ib * Lmax;
            /* rfmini_modrf.py:119-130 */
            double acc = 0.0;
            for (int i = 0; i < n; ++i) {
                zz[i] = acc; /* z = concatenate(([0], cumsum(h)[:-1])) */
                acc += hh[i];
                qp[i] = 500.;
                qs[i] = 225.;
            }
            double vpvs = pvp[0] / pvs[0];
            double poisson = (2 - vpvs * vpvs) / (2 - 2 * (vpvs * vpvs));
            bho_synrf(nsamp, fsamp, tshift, p_s_per_deg, gauss, pvs[0], poisson, waveno, n, zz, pvp,
                      pvs, prh, qp, qs, rf);
            memcpy(rf_out + (size_t)ib * nkeep, rf, sizeof(double) * (size_t)nkeep);
        }
        free(rf);
        free(zz);
    }
}
