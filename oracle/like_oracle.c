/*
 * oracle/like_oracle.c -- CPU restatement of the noise-covariance laws and the Gaussian
 * log-likelihood of src/Targets.py.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Deliberately keeps the reference's O(n^2) formulation -- build the dense inverse
 * covariance (Targets.py:105-173), then  madist = (d^T C^-1) d  (Targets.py:339-342) -- so
 * that it is an independent check of the O(n) closed forms the engine evaluates on the GPU.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle.h"

/* Targets.py:99-103 */
double bho_rms(int n, const double *ymod, const double *yobs)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        double d = ymod[i] - yobs[i];
        s += d * d;
    }
    return sqrt(s / (double)n);
}

double bho_loglike_dense(int law, int n, const double *ymod, const double *yobs,
                         const double *yerr, double corr, double sigma, const double *rinv,
                         double logdet_r)
{
    double *cinv = (double *)calloc((size_t)n * (size_t)n, sizeof(double));
    double logdet = 0.0;
    const double s2 = sigma * sigma;
    switch (law) {
    case BHO_LAW_NOCORR: /* Targets.py:105-115 */
        for (int i = 0; i < n; ++i) cinv[(size_t)i * n + i] = 1.0 / s2;
        logdet = (2.0 * n) * log(sigma);
        break;
    case BHO_LAW_NOCORR_SCALED: { /* Targets.py:117-129: yerr/min(yerr), NOT squared */
        double emin = yerr[0];
        for (int i = 1; i < n; ++i)
            if (yerr[i] < emin) emin = yerr[i];
        double prod = 1.0;
        for (int i = 0; i < n; ++i) {
            double se = yerr[i] / emin;
            cinv[(size_t)i * n + i] = 1.0 / (se * s2);
            prod *= se;
        }
        logdet = (2.0 * n) * log(sigma) + log(prod);
        break;
    }
    case BHO_LAW_EXP: { /* Targets.py:131-148: tridiagonal inverse of r^|i-j| */
        double den = s2 * (1.0 - corr * corr);
        for (int i = 0; i < n; ++i) {
            double dd = 1.0 + corr * corr;
            if (i == 0 || i == n - 1) dd = 1.0;
            cinv[(size_t)i * n + i] = dd / den;
            if (i + 1 < n) {
                cinv[(size_t)i * n + i + 1] = -corr / den;
                cinv[(size_t)(i + 1) * n + i] = -corr / den;
            }
        }
        logdet = (2.0 * n) * log(sigma) + (n - 1) * log(1.0 - corr * corr);
        break;
    }
    case BHO_LAW_GAUSS: /* Targets.py:162-173: fixed R^-1 (host LAPACK, once), scaled */
        for (size_t i = 0; i < (size_t)n * (size_t)n; ++i) cinv[i] = rinv[i] / s2;
        logdet = (2.0 * n) * log(sigma) + logdet_r;
        break;
    default:
        free(cinv);
        return NAN;
    }
    /* (d^T C^-1) d, Targets.py:339-340 */
    double madist = 0.0;
    for (int j = 0; j < n; ++j) {
        double v = 0.0;
        for (int i = 0; i < n; ++i) v += (ymod[i] - yobs[i]) * cinv[(size_t)i * n + j];
        madist += v * (ymod[j] - yobs[j]);
    }
    free(cinv);
    double part = -0.5 * ((double)n * log(2.0 * M_PI) + logdet);
    return part - madist / 2.0;
}

void bho_joint_batch(int B, int Lmax, const int32_t *nlay, const double *h, const double *vp,
                     const double *vs, const double *rho, int nt, const bho_target *targets,
                     const double *noise, double *logL, double *misfits, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    int nmax = 1, nsmax = 1;
    for (int t = 0; t < nt; ++t) {
        if (targets[t].n > nmax) nmax = targets[t].n;
        if (targets[t].kind == 1 && targets[t].nsamp > nsmax) nsmax = targets[t].nsamp;
    }
    if (nsmax > nmax) nmax = nsmax;
#pragma omp parallel
    {
        double *ymod = (double *)malloc(sizeof(double) * (size_t)nmax);
        double *zz = (double *)malloc(sizeof(double) * (size_t)Lmax * 3);
        double *qp = zz + Lmax, *qs = qp + Lmax;
#pragma omp for schedule(dynamic, 4)
        for (int ib = 0; ib < B; ++ib) {
            const int n = nlay[ib];
            const double *hh = h + (size_t)ib * Lmax, *pvp = vp + (size_t)ib * Lmax;
            const double *pvs = vs + (size_t)ib * Lmax, *prh = rho + (size_t)ib * Lmax;
            double ll = 0.0, joint = 0.0;
            int failed = 0;
            for (int t = 0; t < nt && !failed; ++t) {
                const bho_target *T = &targets[t];
                if (T->kind == 0) {
                    float fh[100], fvp[100], fvs[100], frho[100];
                    for (int i = 0; i < n; ++i) {
                        fh[i] = (float)hh[i]; fvp[i] = (float)pvp[i]; fvs[i] = (float)pvs[i]; frho[i] = (float)prh[i];
                    }
                    if (bho_surfdisp96(fh, fvp, fvs, frho, n, 0, T->iwave, 1, T->igr, T->n, T->x, ymod, NULL)) failed = 1;
                } else {
                    double acc = 0.0;
                    for (int i = 0; i < n; ++i) { zz[i] = acc; acc += hh[i]; qp[i] = 500.; qs[i] = 225.; }
                    double k = pvp[0] / pvs[0];
                    double poisson = (2 - k * k) / (2 - 2 * (k * k));
                    bho_synrf(T->nsamp, T->fsamp, T->tshift, T->p_s_per_deg, T->gauss, pvs[0], poisson,
                              T->waveno, n, zz, pvp, pvs, prh, qp, qs, ymod);
                }
                if (failed) break;
                const double corr = noise[(size_t)ib * 2 * nt + 2 * t], sigma = noise[(size_t)ib * 2 * nt + 2 * t + 1];
                ll += bho_loglike_dense(T->law, T->n, ymod, T->yobs, T->yerr, corr, sigma, NULL, 0.0);
                double r = bho_rms(T->n, ymod, T->yobs);
                misfits[(size_t)ib * (nt + 1) + t] = r;
                joint += r;
            }
            if (failed) { /* Targets.py:325-328 */
                logL[ib] = -1e15;
                for (int t = 0; t <= nt; ++t) misfits[(size_t)ib * (nt + 1) + t] = 1e15;
            } else {
                logL[ib] = ll;
                misfits[(size_t)ib * (nt + 1) + nt] = joint;
            }
        }
        free(ymod);
        free(zz);
    }
}

/* The host libm itself, vectorised: what the reference's Fortran / C++ call (sincos, exp).
 * Used by the tests to show that the device's restatement (csrc/bh_libm.h) returns the same bits. */
void bho_libm_probe(int op, int n, const double *in, double *out)
{
    /* through volatile pointers: the compiler must really call sincos() (what the reference's
     * compiled Fortran calls) and not narrow it to sin()/cos(), which are different (FMA ifunc)
     * implementations in glibc and differ from sincos() in the last bit for ~0.1 % of arguments */
    void (*volatile p_sincos)(double, double *, double *) = sincos;
    double (*volatile p_exp)(double) = exp;
    for (int i = 0; i < n; ++i) {
        double s, c;
        switch (op) {
        case 0: p_sincos(in[i], &s, &c); out[i] = s; break;
        case 1: p_sincos(in[i], &s, &c); out[i] = c; break;
        default: out[i] = p_exp(in[i]); break;
        }
    }
}
