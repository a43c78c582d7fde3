/*
 * oracle/like_oracle.c -- CPU restatement of the noise-covariance laws and the Gaussian
 * log-likelihood of src/Targets.py.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Deliberately keeps the reference's O(n^2) formulation -- build the dense inverse
 * covariance (Targets.py:105-173), then  madist = (d^T C^-1) d  (Targets.py:339-342) -- so
 * that it is an independent check of the O(n) closed forms the engine evaluates on the GPU.
 */
#include <math.h>
#include <stdlib.h>
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
