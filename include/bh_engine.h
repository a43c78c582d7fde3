/*
 * include/bh_engine.h -- C ABI of the MI355X forward-model + likelihood engine.
 *
 * This is the drop-in boundary: the entry points below are what BayHunter's two native
 * FFI surfaces for the hot path bind to, batched over many candidate models.
 *
 *   reference interface replaced                                    entry point here
 *   ----------------------------------------------------------      -----------------
 *   f2py  surfdisp96(thkm,vpm,vsm,rhom,nlayer,iflsph,iwave,mode,    bh_swd_batch
 *         igr,kmax,t,cg) -> err      src/extensions/surfdisp96.f:55
 *         (call site src/surf96_modsw.py:115-117)
 *   C     int synrf_cwrap(nsamp,fsamp,tshift,p,a,nsv,sigma,waveno,  bh_rf_batch
 *         nlay,z,vp,vs,rh,qp,qs,fz,fr,rf)
 *         src/extensions/rfmini/wrap.cpp:58-80
 *         (call site src/rfmini_modrf.py:134-137 via rfmini.pyx)
 *   Python JointTarget.evaluate(h,vp,vs,noise)  src/Targets.py:314-347   bh_targets_set +
 *         + Valuation.get_covariance_* / get_likelihood :105-183    bh_evaluate_batch
 *
 * Conventions
 *   - Plain C: pointers + sizes, no C++/torch types.  Every function returns BH_OK (0) or a
 *     negative BH_E* code; bh_engine_last_error() gives the text.  A NUMERICAL failure of one
 *     model (surf96 finds no root) is reported IN-BAND per model -- err[b] = 1, velocities 0
 *     from the failing period on, logL = -1e15, misfits = 1e15 -- exactly like the reference
 *     (surfdisp96.f:313-354, surf96_modsw.py:119-126, Targets.py:325-328), never as a
 *     non-zero return.
 *   - memspace = BH_HOST: every array argument is a host pointer; the call stages through
 *     engine-owned device buffers and returns after the results are back on the host.
 *     memspace = BH_DEVICE: every array argument is a device pointer on the engine's GPU
 *     (e.g. torch tensor .data_ptr()); the call only enqueues work on `stream` (a hipStream_t
 *     cast to void*; NULL = the engine's own stream) and returns without synchronising.
 *   - Model arrays h, vp, vs, rho are float64 with element (layer l, model b) at
 *     ptr[l*stride_l + b*stride_b].  Layer-major storage (stride_l = B, stride_b = 1) gives
 *     coalesced loads on the device; model-major rows (stride_l = 1, stride_b = Lmax) are what
 *     a host caller naturally has.  nlay[b] <= Lmax is the number of layers of model b
 *     INCLUDING the half-space (whose thickness entry is ignored); entries l >= nlay[b] are
 *     never read.
 *   - The caller owns every buffer it passes.  The engine keeps no pointer after a call
 *     returns, except copies of the constant target data registered by bh_targets_set.
 *   - Measurement, diagnostic and experiment entry points (timing, counters, launch-geometry knobs, probes) live in
 *     include/bh_engine_debug.h: not part of this contract.
 *   - One engine = one GPU = one stream; calls on one engine must be serialised by the caller
 *     (the reference is single-threaded per chain as well).
 */
#ifndef BH_ENGINE_H
#define BH_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BH_ABI_VERSION 10

enum { BH_OK = 0, BH_EINVAL = -1, BH_EHIP = -2, BH_ENOMEM = -3, BH_EUNSUPPORTED = -4 };
enum { BH_HOST = 0, BH_DEVICE = 1 };

/* surfdisp96's `iwave` (surfdisp96.f:76) and `igr` (:78) */
enum { BH_WAVE_LOVE = 1, BH_WAVE_RAYLEIGH = 2 };
enum { BH_VEL_PHASE = 0, BH_VEL_GROUP = 1 };
/* rfmini's `waveno` (rfmini/wave.h:4-6) */
enum { BH_RF_P = 0, BH_RF_SV = 1 };

/* surfdisp96.f:59-62 hard limits, kept so that anything the reference accepts is accepted */
#define BH_MAX_LAYERS 100
#define BH_MAX_PERIODS 60
#define BH_MAX_TARGETS 8

typedef struct bh_engine bh_engine;

int bh_abi_version(void);

/* Create an engine on GPU `device` (hipSetDevice ordinal).  Fails with BH_EHIP when no
 * usable gfx950 device is present: there is no CPU fallback in this library. */
int bh_engine_create(int device, bh_engine **out);
void bh_engine_destroy(bh_engine *e);
const char *bh_engine_last_error(const bh_engine *e);
/* Root refinement of the dispersion search (surfdisp96.f:390-686).
 *   BH_SEARCH_REFERENCE  the reference's sequence of secular-function evaluations (getsol + nevill), evaluation for
 *                        evaluation: velocities and failure flags bit-identical to the reference's.
 *   BH_SEARCH_FAST       (the default) the reference's bracket scan -- the same grid, the same bracket -- and inside the
 *                        bracket a few evaluations instead of nevill's ten to twelve.  GUARANTEES: velocities within 1e-5
 *                        relative of the reference's (asserted: 2e-6); the failure flag and the period from which a failed
 *                        model's row is zero are the reference's.  A model whose outcome could hinge on the last digits of a
 *                        root, or on WHICH of several roots of one scan cell nevill ends at (a cell that holds a half-space
 *                        velocity or betmx can hold a root, its mirror image and more), is detected and run again with the
 *                        reference's sequence inside the same call; bh_engine_guard_stats counts them.  (Known exception,
 *                        fewer than one in a million models drawn from a sampler's prior, none seen on models with sorted
 *                        velocities: a root within ~1e-6 c of a scan grid point with a second root less than a step away, or
 *                        two sign changes of a cell with a half-space velocity closer to each other than the count's points
 *                        (a distance ratio of 1.23 from that velocity) -- the scan can end on another mode than the reference's,
 *                        and fail where it does not or the other way round further along that branch.  DESIGN.md 4.)  NOT the reference's
 *                        bits.  WHAT A RESULT DEPENDS ON: the model, and -- in its last digits, ~1e-9 relative, up to the
 *                        reference's own 1e-6 where the guard fires under one setting and not under another -- the kernel that
 *                        ran and its trials per round.  With BH_ARITH_EXACT that is the model alone.  With BH_ARITH_FAST
 *                        (the default) the trial-per-lane kernel picks its trials per round from the call's shape (64 up to
 *                        1024 (model, target) pairs, 32 up to 5120, 16 up to 10240, 8 up to 28672, 4 beyond; arrays deeper
 *                        than 32 layers, or whose LDS need exceeds a workgroup's, take the lane-per-evaluation kernel): the SAME
 *                        model in calls of different shapes may differ in those last digits.  bh_engine_set_swd_trials pins the
 *                        number: then a model's result does not depend on the batch it is in, its position, or the launch
 *                        (tests/test_gpu_swd_lean.py::test_result_is_a_function_of_the_model_and_the_trials).  Applies to
 *                        fundamental-mode phase-velocity targets; group-velocity targets and targets with higher modes always
 *                        take the reference's sequence.
 *   BH_SEARCH_FAST_RAYLEIGH  BH_SEARCH_FAST for Rayleigh targets, BH_SEARCH_REFERENCE for Love targets.
 * A replay against chains recorded with the reference needs BH_SEARCH_REFERENCE. */
#define BH_SEARCH_REFERENCE 0
#define BH_SEARCH_FAST 1
#define BH_SEARCH_FAST_RAYLEIGH 2
int bh_engine_set_swd_search(bh_engine *e, int search);
int bh_engine_get_swd_search(const bh_engine *e);
/* Arithmetic of the secular functions in launches where EVERY target takes the short refinement (BH_SEARCH_FAST with
 * fundamental-mode phase-velocity targets only; anything else -- BH_SEARCH_REFERENCE, group velocities, higher modes, the
 * re-run of guarded models -- always computes with BH_ARITH_EXACT).
 *   BH_ARITH_EXACT  the reference's operations with the reference's rounding points (no fused multiply-add, correctly rounded
 *                   division and square root, glibc's sincos / exp bit for bit): what the short refinement evaluates is the
 *                   reference's function to the last bit.
 *   BH_ARITH_FAST   (the default) the same formulas with fused multiply-adds, Newton-refined hardware reciprocals / square
 *                   roots and short polynomial sin / cos / exp: values within a few units in the last place of the exact
 *                   ones (|f_fast - f_exact| <= 1.2e-8 of the vector's max-norm, below 1e-11 in 99.6 % of 10^8 sampled
 *                   evaluations) -- a root moves by ~1e-13 relative; a scan's sign pattern can differ from the exact one's
 *                   only where a root lies within ~1e-8 of a grid point (the bracket then moves one step around the same
 *                   root).  The GUARANTEES of BH_SEARCH_FAST (tolerance, failure flags, zero rows) are asserted for this
 *                   arithmetic on 1.7 million fixed LVZ-rich models and, in every run of the suite, on 200 000 fresh ones
 *                   incl. models drawn from a sampler's prior (tests/test_gpu_fuzz.py); a value that is not a number (an
 *                   argument beyond the reduction's range) sends the model back to the reference's sequence and arithmetic,
 *                   like any guarded model. */
#define BH_ARITH_EXACT 0
#define BH_ARITH_FAST 1
int bh_engine_set_swd_arith(bh_engine *e, int arith);
int bh_engine_get_swd_arith(const bh_engine *e);
/* Trials per model and round of the trial-per-lane kernel (BH_SEARCH_FAST + BH_ARITH_FAST): 0 (default) = by the call's shape
 * as described at BH_SEARCH_FAST; 4, 8, 16, 32 or 64 = that many in every call -- what a sampler sets whose windows, initial
 * state and shards must give the same bits whatever their size (DeviceChains: 32). */
int bh_engine_set_swd_trials(bh_engine *e, int trials);
int bh_engine_get_swd_trials(const bh_engine *e);
/* BH_SEARCH_FAST statistics: counts[t] (BH_MAX_TARGETS entries; may be NULL) = models of target t of the most recent
 * dispersion call that its guard sent back to the reference's sequence (listed for the re-run launch, or restarted in place
 * in a launch of one model per wavefront); *rerun_launches (may be NULL) = re-run launches
 * enqueued since the engine was created; total[t] (BH_MAX_TARGETS entries; may be NULL) = guarded models of the t-th dispersion
 * target of the calls since the engine was created.
 * Synchronises the engine's stream when counts or total is given. */
int bh_engine_guard_stats(bh_engine *e, int32_t *counts, uint64_t *rerun_launches, uint64_t *total);
/* Tuning hint for BH_DEVICE calls: the typical number of layers (incl. the half-space) of the models in
 * the batches to come, when it is well below Lmax (transdimensional chains: capacity 21, typically 5-7).
 * The lanes-per-model choice is sized for it; 0 = unknown (Lmax is used).  BH_HOST calls look at nlay
 * themselves.  Results do not depend on it. */
int bh_engine_set_typical_layers(bh_engine *e, int nlay);
/* Scheduling hint: by default a call first sorts its batch by layer count on the device (a counting sort, ~12 us + a
 * launch gap) so that a wavefront holds models of one depth -- what ragged (transdimensional) batches need.  A caller
 * whose batches are of uniform depth (or already sorted) turns it off with sort_by_depth = 0.  Results do not depend on it. */
int bh_engine_set_model_order(bh_engine *e, int sort_by_depth);
/* The engine's own stream as a hipStream_t cast to void*. */
void *bh_engine_stream(bh_engine *e);
/* Block until everything enqueued on the engine's stream has finished. */
int bh_engine_synchronize(bh_engine *e);

/* ---- surface-wave dispersion: replaces surfdisp96 (surfdisp96.f:55-360) -----------------
 * For each of B models: velocities at K <= 60 periods [s] for wave type `iwave`, velocity
 * type `igr`, modes 1..`mode` computed in turn with the values of the last one returned
 * (surfdisp96.f:219-357; a higher mode that finds no root leaves zeros and does not set err,
 * :313), flat (`flsph` = 0) or earth-flattened (`flsph` = 1, surfdisp96.f:486-553) model.
 * Results are bit-identical to the reference, with or without the flattening transform.
 *   vel[b*K + k]  float64, values are binary32-rounded like the reference's output
 *   err[b]        0 ok / 1 no root found (then vel[b][k..] = 0 from the failing period on)
 * A model with a non-finite or non-physical parameter (vp, rho <= 0, vs < 0, velocities > 100 km/s,
 * negative thickness) is reported as failed without being searched: the reference's search loops are
 * bounded only through the model's velocities and do not terminate on such input.
 */
int bh_swd_batch(bh_engine *e, int memspace, void *stream, int B, int Lmax, const int32_t *nlay,
                 const double *h, const double *vp, const double *vs, const double *rho,
                 ptrdiff_t stride_l, ptrdiff_t stride_b, int K, const double *periods, int iwave,
                 int igr, int mode, int flsph, double *vel, int32_t *err);

/* ---- receiver function: replaces synrf_cwrap (rfmini/wrap.cpp:58-80) ----------------------
 * For each of B models: the (Q-component) receiver function for incident `waveno`, ray
 * parameter p [s/deg], Gauss parameter `gauss`, nsamp (power of two, 4 ... 262144; BH_EUNSUPPORTED above) samples at fsamp [Hz],
 * time origin shifted by tshift [s]; the first nkeep <= nsamp samples are returned, which is
 * what rfmini_modrf.py:142 keeps.  Depths of layer tops are cumsum(h) as in
 * rfmini_modrf.py:119-123.  qp/qs: per-layer quality factors with the model-array strides, or
 * NULL for the reference defaults 500/225 (rfmini_modrf.py:116-117).  nsv <= 0 selects the
 * reference default nsv = vs[0] with Poisson's ratio from vp[0]/vs[0]
 * (rfmini_modrf.py:125-130); nsv > 0 is used as given with the same Poisson ratio.
 *   rf[b*nkeep + i] float64
 */
int bh_rf_batch(bh_engine *e, int memspace, void *stream, int B, int Lmax, const int32_t *nlay,
                const double *h, const double *vp, const double *vs, const double *rho,
                const double *qp, const double *qs, ptrdiff_t stride_l, ptrdiff_t stride_b,
                double p_s_per_deg, double gauss, int nsamp, double fsamp, double tshift,
                double nsv, int waveno, int nkeep, double *rf);

/* ---- fused forward model + likelihood: replaces JointTarget.evaluate -----------------------
 * Noise-covariance laws of src/Targets.py:105-173, selected per target the way
 * SingleChain.set_target_covariance does (src/SingleChain.py:159-205). */
enum {
    BH_LAW_NOCORR = 0,        /* Targets.py:105-115 */
    BH_LAW_NOCORR_SCALED = 1, /* Targets.py:117-129, needs yerr */
    BH_LAW_EXP = 2,           /* Targets.py:131-148 */
    BH_LAW_GAUSS = 3          /* Targets.py:150-173, needs rinv (n x n, row-major) + logdet_r */
};
/* BH_TARGET_USER: observed data + noise law only, no forward model in the engine (synthetics
 * come from a user plugin through bh_loglike_batch; bh_evaluate_batch refuses such a set). */
enum { BH_TARGET_SWD = 0, BH_TARGET_RF = 1, BH_TARGET_USER = 2 };

typedef struct bh_target_desc {
    int32_t kind; /* BH_TARGET_SWD | BH_TARGET_RF | BH_TARGET_USER */
    int32_t law;  /* BH_LAW_* */
    int32_t n;    /* number of observed samples (periods or time samples) */
    /* SWD (ignored for RF): */
    int32_t iwave, igr, mode, flsph;
    /* RF (ignored for SWD): */
    int32_t waveno, nsamp;
    double p_s_per_deg, gauss, fsamp, tshift, nsv;
    /* observed data, HOST pointers, copied by bh_targets_set: */
    const double *x;    /* [n] periods [s] (SWD; n > 60: forward model on linspace(min, max, 60) and
                           np.interp back, like surf96_modsw.py:35-43,:119-122) -- for RF the time axis
                           is implied by fsamp/tshift */
    const double *yobs; /* [n] */
    const double *yerr; /* [n] or NULL (only BH_LAW_NOCORR_SCALED reads it) */
    const double *rinv; /* [n*n] or NULL (only BH_LAW_GAUSS reads it) */
    double logdet_r;    /* ln|R| for BH_LAW_GAUSS */
} bh_target_desc;

/* Register nt <= BH_MAX_TARGETS targets (constant data is copied to the device). */
int bh_targets_set(bh_engine *e, int nt, const bh_target_desc *targets);

/* For each of B models: run every registered target's forward model, then
 *   logL[b]            = sum_t -1/2 (n_t ln 2pi + ln|C_t|) - 1/2 d_t^T C_t^-1 d_t   (Targets.py:339-344)
 *   misfits[b*(nt+1)+t]= RMS_t, last entry their sum                         (Targets.py:307-312)
 *   err[b]             = 1 if any target's forward model failed (then logL = -1e15, misfits = 1e15)
 * noise[b*2*nt + 2*t + {0,1}] = (corr, sigma) of target t (Targets.py:335).
 * rho may be NULL: rho = 0.32*vp + 0.77 (Targets.py:319).
 * ymod (optional, may be NULL): the synthetic data, target after target, [b][sum_t n_t].
 */
int bh_evaluate_batch(bh_engine *e, int memspace, void *stream, int B, int Lmax,
                      const int32_t *nlay, const double *h, const double *vp, const double *vs,
                      const double *rho, ptrdiff_t stride_l, ptrdiff_t stride_b,
                      const double *noise, double *logL, double *misfits, int32_t *err,
                      double *ymod);

/* Likelihood only, on synthetics the caller already has (e.g. produced by a user-supplied
 * forward-modelling plugin, Targets.py:201-202 `update_plugin`): same outputs as
 * bh_evaluate_batch, with ymod[b][sum_t n_t] given and fail[t*B + b] != 0 marking a failed
 * forward model of target t (may be NULL = none failed). */
int bh_loglike_batch(bh_engine *e, int memspace, void *stream, int B, const double *ymod,
                     const int32_t *fail, const double *noise, double *logL, double *misfits,
                     int32_t *err);

/* ---- device-resident chain step: replaces the host part of SingleChain.iterate --------------
 * C chains advance in lock-step, one iteration = bh_chain_propose -> bh_evaluate_batch (BH_DEVICE,
 * on state->lay_*, stride_l = C, stride_b = 1, noise = state->pnoise) -> bh_chain_accept.
 * bh_chain_propose: modification choice, proposal, nuclei sort, prior/validity checks and the
 *   Voronoi->layer conversion (src/SingleChain.py:246-420, :511-556; src/Models.py:26-52).
 * bh_chain_accept: acceptance incl. the birth/death terms, state update, counters and the
 *   proposal-width adaptation every 1000 iterations (src/SingleChain.py:425-487, :558-589).
 * All arrays are DEVICE pointers, float64 unless noted; "[k][C]" = k rows of C chains (chain index
 * contiguous).  Random numbers are Philox4x32-10 keyed by `seed` with counter (global chain index, iteration,
 * purpose): reproducible and independent of scheduling, but a different stream from the
 * reference's per-chain Mersenne Twister -- chains agree with the reference statistically (the
 * draw-for-draw replay is the host driver bayhunter_amd/chains.py).  If state->inject is not
 * NULL the six draws of an iteration are read from it instead ([6][C]: u_move, u_index, u_z,
 * u_accept, u_noise in [0,1) and one standard normal) -- used by the tests. */
#define BH_CHAIN_MAXLAYERS 32 /* upper bound of cfg.maxlayers (nuclei per chain) */
#define BH_CHAIN_MAXDEPTH 7   /* iterations per speculative window (2^7 - 1 = 127 proposals per chain) */

typedef struct bh_chain_config {
    int32_t nt;                   /* targets (noise has 2*nt entries: corr, sigma per target) */
    int32_t maxlayers;            /* row capacity of the nuclei arrays = priors 'layers' max + 1 */
    int32_t layermin, layermax;   /* priors 'layers' (number of layers above the half space) */
    int32_t iter_burnin, iterations;
    double vsmin, vsmax, zmin, zmax; /* priors 'vs', 'z' */
    double thickmin;              /* initparams 'thickmin' */
    double lvz, hvz;              /* initparams 'lvz' / 'hvz'; < 0 = None */
    double vpvsmin, vpvsmax;      /* priors 'vpvs'; equal = fixed */
    double mantle_vs, mantle_vpvs; /* priors 'mantle' (vs threshold, vp/vs below); mantle_vs <= 0 = None */
    double acc_lo, acc_hi;        /* initparams 'acceptance' [%] */
    double noise_lo[2 * BH_MAX_TARGETS], noise_hi[2 * BH_MAX_TARGETS]; /* equal = fixed */
    uint64_t seed;
    int64_t chain_offset;         /* global index of this call's chain 0 (sharded jobs: the Philox counter
                                     uses chain_offset + c, so that one job-wide seed gives every chain its own stream) */
} bh_chain_config;

typedef struct bh_chain_state {
    /* current state */
    int32_t *n;       /* [C] nuclei */
    double *vs, *z;   /* [maxlayers][C] nuclei, sorted by depth */
    double *vpvs;     /* [C] */
    double *noise;    /* [2nt][C] */
    double *like;     /* [C] log-likelihood of the current model */
    double *misfits;  /* [nt+1][C] */
    double *propdist; /* [5][C] vs, z, birth/death, noise, vpvs (SingleChain.py:117) */
    double *proposed, *accepted; /* [5][C] counters */
    int64_t *naccepted; /* [C] */
    const double *beta; /* NULL or [C]: inverse temperature of each chain (parallel tempering: the
                           likelihood ratio enters the acceptance as beta*(logL' - logL); 1 = the reference) */
    /* proposal (written by bh_chain_propose, read by bh_chain_accept) */
    int32_t *pn, *move, *valid; /* [C] */
    double *pvs, *pz;  /* [maxlayers][C] */
    double *pvpvs;     /* [C] */
    double *pnoise;    /* [C][2nt] -- bh_evaluate_batch's layout */
    double *dvs2;      /* [C] */
    /* layered model of the proposal (of the current model where the proposal is invalid) */
    int32_t *lay_n;    /* [C] layers incl. half space */
    double *lay_h, *lay_vp, *lay_vs; /* [maxlayers][C] */
    const double *inject; /* NULL or [6][C] */
    double *lay_rho;   /* NULL or [maxlayers][C]: density of the proposal's layers, 0.32 vp + 0.77 (Targets.py:319), so that
                          bh_evaluate_batch need not derive it in a launch of its own */
} bh_chain_state;

int bh_chain_propose(void *stream, const bh_chain_config *cfg, const bh_chain_state *state, int C, int iiter);
/* logL [C], misfits [C][nt+1]: outputs of bh_evaluate_batch for this iteration (device). */
int bh_chain_accept(void *stream, const bh_chain_config *cfg, const bh_chain_state *state, int C, int iiter,
                    const double *logL, const double *misfits);

/* Speculative window: `depth` (1..BH_CHAIN_MAXDEPTH) iterations iiter .. iiter+depth-1 of every chain per evaluation
 * launch, with exactly the results of `depth` propose/evaluate/accept rounds (SingleChain.py:511-589 is a sequential
 * loop; its draws here are a pure function of (chain, iteration), so the proposals of both outcomes of every
 * decision can be written down in advance).  The proposal members of `state` (pn, move, valid, pvs, pz, pvpvs,
 * pnoise, dvs2, lay_*) then hold N = 2^depth - 1 proposals per chain in heap order -- node 0 = the proposal of
 * iteration iiter, nodes 2j+1 / 2j+2 = the proposals of the next iteration after node j was rejected / accepted --
 * node j of chain c in column j*C + c of arrays with `ld` >= N*C columns ("[k][ld]" instead of "[k][C]"; pnoise
 * [ld][2nt]).  One window = bh_chain_propose_window -> bh_evaluate_batch(B = N*C, stride_l = ld, stride_b = 1,
 * noise = pnoise) -> bh_chain_accept_window(logL [N*C], misfits [N*C][nt+1]), which walks the realised path.
 * An iteration with iiter % 1000 == 0 (proposal-width adaptation) must be the last of its window (BH_EINVAL
 * otherwise); state->inject, if used, holds [depth][6][C].  depth = 1, ld = C is bh_chain_propose / _accept. */
int bh_chain_propose_window(void *stream, const bh_chain_config *cfg, const bh_chain_state *state, int C, int iiter,
                            int depth, ptrdiff_t ld);
int bh_chain_accept_window(void *stream, const bh_chain_config *cfg, const bh_chain_state *state, int C, int iiter,
                           int depth, ptrdiff_t ld, const double *logL, const double *misfits);

#ifdef __cplusplus
}
#endif
#endif
