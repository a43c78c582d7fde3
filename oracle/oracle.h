/*
 * oracle/oracle.h -- CPU restatement of the BayHunter forward-model + likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bayhunter_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker / reported CPU baseline.
 *
 * Parity status: PINNED.  liboracle.so is checked (tests/test_oracle_*.py) against
 *   (a) the reference's own golden files tutorial/observed/st3_*.dat (copied as data into
 *       tests/golden/st3/), and
 *   (b) tests/golden/ (.npz files), produced in the build container by tests/golden/gen_golden.py
 *       from the UNMODIFIED reference sources compiled by `make -C oracle ref`
 *       (surfdisp96.f via amdflang, the rfmini .cpp files via g++) and the reference's Python layer.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * root, i.e. jenndrei/BayHunter v2.1).
 */
#ifndef BH_ORACLE_H
#define BH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- surf96 (src/extensions/surfdisp96.f) ------------------------------------------ */

/* Whole driver, src/extensions/surfdisp96.f:55-360.  Model arrays are binary32 exactly as
 * the f2py wrapper hands them over (surf96_modsw.py:115-117).  Returns `err` (0 ok, 1 no
 * root in the fundamental mode).  cg[0..kmax) receives the velocities.  If neval != NULL it
 * receives the number of secular-function evaluations spent (for the flop model). */
int bho_surfdisp96(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                   int nlayer, int iflsph, int iwave, int mode, int igr, int kmax,
                   const double *t, double *cg, int64_t *neval);

/* 1: bho_surfdisp96 / bho_swd_batch refine a bracketed root by the engine's optional short sequence (phase velocities
 * only) instead of the reference's nevill -- a restatement of THIS repo's swd_common.h, not of the reference; 0 (default):
 * the reference's sequence.  Process-wide switch. */
void bho_swd_set_search(int fast);
/* 2: the same, GUARDED the way the engine runs BH_SEARCH_FAST: a model whose short-sequence run fails in any mode or accepts
 * a root within two scan steps of the fastest S velocity / a half-space velocity is run again with the reference's sequence
 * (swd_oracle.c, bho_surfdisp96).  bho_swd_guarded_count: how many models that was since the last reset. */
int64_t bho_swd_guarded_count(int reset);
/* Scan mode: 0 (default here) = getsol's scan, one step of dc per evaluation -- the restatement proper, the one pinned to the
 * compiled reference; 1 = the engine's counted scan (Love: steps a mode count proves to be without a sign change are not
 * visited; swd_oracle.c, bracket_and_refine): the same brackets and bits with fewer evaluations, restated here so that the
 * device's evaluation counts can be checked and the certificate itself tested against mode 0 on the CPU. */
void bho_swd_set_scan(int counted);
/* 1: fundamental-mode group velocities as the device runs them -- the chain of the first roots, then the second roots on their own
 * (swd_oracle.c, g_group_split): a different ORDER of the reference's searches, checked against the reference's on the CPU. */
void bho_swd_set_group_split(int on);
void bho_swd_set_scan_tuning(int first, int next, int back);
double bho_dltar1_count(double wvno, double omega, const float *d, const float *b, const float *rho,
                        int mmax, int llw, int *count, int *valid);

/* Secular functions, exposed so that tests can compare them 1:1 with the reference's
 * exported dltar1_/dltar4_ symbols.  surfdisp96.f:710-769 and :773-871. */
double bho_dltar1(double wvno, double omega, const float *d, const float *b,
                  const float *rho, int mmax, int llw);
double bho_dltar4(double wvno, double omega, const float *d, const float *a, const float *b,
                  const float *rho, int mmax, int llw);
/* surfdisp96.f:367-388 */
float bho_gtsolh(float a, float b);

/* the surface vector of the binary64 recursion (2 / 5 entries) */
void bho_secular_vec(int ifunc, double omega, double c, const float *d, const float *a, const float *b, const float *rho,
                     int mmax, int llw, double *vec);

/* Batched convenience: B models, SoA-by-model-row [B][Lmax] float64 inputs (cast to f32
 * inside, like f2py does), OpenMP over models.  vel[B][K], err[B]. */
void bho_swd_batch(int B, int Lmax, const int32_t *nlay, const double *h, const double *vp,
                   const double *vs, const double *rho, int K, const double *periods,
                   int iwave, int igr, int mode, int flsph, double *vel, int32_t *err,
                   int64_t *neval_total, int nthreads);

/* ---- rfmini (src/extensions/rfmini) -------------------------------------------------- */

/* extern "C" synrf_cwrap of rfmini/wrap.cpp:58-80 -> synrf.cpp:16-55 -> greens.cpp:685-756,
 * receiver function only (the z/r traces are computed and discarded by the Python caller,
 * rfmini_modrf.py:134-142).  rf[nsamp].  Returns 1 like the reference. */
int bho_synrf(int nsamp, double fsamp, double tshift, double p, double a, double nsv,
              double sigma, int waveno, int nlay, const double *z, const double *vp,
              const double *vs, const double *rh, const double *qp, const double *qs,
              double *rf);

/* Batched: h given as thickness (last = 0), z built as rfmini_modrf.py:119-123 does;
 * qp/qs = 500/225; nsv = vs[0], sigma from vp[0]/vs[0] (rfmini_modrf.py:125-130).
 * rf_out[B][nkeep]. */
void bho_rf_batch(int B, int Lmax, const int32_t *nlay, const double *h, const double *vp,
                  const double *vs, const double *rho, double p_s_per_deg, double gauss,
                  int nsamp, double fsamp, double tshift, int waveno, int nkeep,
                  double *rf_out, int nthreads);

/* ---- likelihood (src/Targets.py) ----------------------------------------------------- */

enum { BHO_LAW_NOCORR = 0, BHO_LAW_NOCORR_SCALED = 1, BHO_LAW_EXP = 2, BHO_LAW_GAUSS = 3 };

/* Targets.py:339-342 with the covariance laws of :105-173, evaluated the way the reference
 * does it: build the dense n x n inverse covariance and do (d^T C^-1) d.  O(n^2) on purpose --
 * this is the checker for the closed forms the engine uses.  rinv/logdet_r only for
 * BHO_LAW_GAUSS (R^-1 and ln|R| come from host LAPACK in the reference, Targets.py:150-160). */
double bho_loglike_dense(int law, int n, const double *ymod, const double *yobs,
                         const double *yerr, double corr, double sigma, const double *rinv,
                         double logdet_r);

/* Targets.py:99-103 */
double bho_rms(int n, const double *ymod, const double *yobs);

/* JointTarget.evaluate (Targets.py:314-347) for B models, OpenMP over models: per target the
 * forward model above + the dense likelihood.  This is bench.py's cpu_baseline ("port"). */
typedef struct bho_target {
    int32_t kind;  /* 0 SWD, 1 RF */
    int32_t law;   /* BHO_LAW_* (Gauss law not supported here) */
    int32_t n;
    int32_t iwave, igr;              /* SWD */
    int32_t waveno, nsamp;           /* RF */
    double p_s_per_deg, gauss, fsamp, tshift;
    const double *x;                 /* [n] periods (SWD) */
    const double *yobs;              /* [n] */
    const double *yerr;              /* [n] or NULL */
} bho_target;
void bho_joint_batch(int B, int Lmax, const int32_t *nlay, const double *h, const double *vp,
                     const double *vs, const double *rho, int nt, const bho_target *targets,
                     const double *noise, double *logL, double *misfits, int nthreads);

/* host libm (sincos / exp) on arrays: op 0 sin, 1 cos (both through sincos()), 2 exp */
void bho_libm_probe(int op, int n, const double *in, double *out);

#ifdef __cplusplus
}
#endif
#endif
