/*
 * include/bh_engine_debug.h -- measurement, diagnostic and experiment entry points of libbh_engine.so.
 *
 * NOT part of the drop-in contract (include/bh_engine.h): nothing here changes a result a caller of bh_engine.h sees, and a
 * BayHunter binding needs none of it.  bench.py, tools/ and the tests use it: launch-geometry knobs of the layer-parallel
 * dispersion kernel, the experiment switches (csrc/bh_tuning.h), HIP-event timing and evaluation counters, probes of the device
 * math library.  May change between ABI versions without notice.
 */
#ifndef BH_ENGINE_DEBUG_H
#define BH_ENGINE_DEBUG_H

#include "bh_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning knob: lanes of a wavefront that cooperate on ONE model in the dispersion kernel
 * (1..32; 0 = choose from the batch size and layer count, the default).  Results do not depend on it. */
int bh_engine_set_swd_group(bh_engine *e, int lanes_per_model);
/* Tuning knob: trial phase velocities evaluated per round of the root search in the dispersion
 * kernel (1..16; 0 = choose from the batch size, the default; with one lane per model -- group 1 -- the
 * largest power of two not above it is used).  The search (surfdisp96.f:390-686) asks
 * for one secular-function value at a time; with look-ahead the kernel also evaluates, on further
 * lanes, the velocities the search will most probably ask for next and hands them over if and only
 * if it does.  Results do not depend on it. */
int bh_engine_set_swd_lookahead(bh_engine *e, int trials_per_round);

/* Which dispersion kernel the most recent call with dispersion targets launched (diagnostic; -1: none yet):
 *   BH_KERNEL_GROUP  several lanes per model, layer-parallel (swd_group_kernel),
 *   BH_KERNEL_LANE   one lane per evaluation, reference-exact or FA builds (swd_kernel),
 *   BH_KERNEL_LEAN   one lane per trial velocity with the fast arithmetic (swd_lean_kernel): BH_SEARCH_FAST + BH_ARITH_FAST
 *                    calls whose targets are all fundamental-mode phase velocities, in arrays of up to 32 layers.  It
 *                    evaluates 64 trial velocities per model and round in calls of up to 1024 (model, target) pairs, 32 up
 *                    to 5120, 16 up to 10240, 8 up to 28672 and 4 beyond; a model's velocities depend on that number in their last bits (~1e-9 relative; a
 *                    model whose guard fires under one count and not under another: the reference's 1e-6) and on nothing
 *                    else about the call. */
#define BH_KERNEL_GROUP 0
#define BH_KERNEL_LANE 1
#define BH_KERNEL_LEAN 2
int bh_engine_last_swd_kernel(const bh_engine *e);

/* The bracket scan of Love targets (any root refinement).  Results never depend on this setting.
 * getsol's scan (surfdisp96.f:437-460) evaluates every step of its grid until the secular function changes sign.  For Love
 * waves the number of sign changes below a trial velocity is read off the recursion that evaluates the function (a Sturm
 * count), so two evaluations prove that the steps between them hold no sign change (they are skipped) or exactly one (it is
 * located by a search over the step index).  Grid points, bracket and every bit after it are the reference's.
 *   BH_SCAN_STEPS    every step evaluated, as the reference does.
 *   BH_SCAN_COUNTED  the counted scan wherever a launch holds a Love target -- except launches of several models per
 *                    wavefront that mix both root refinements (that kernel build does not carry it).
 *   BH_SCAN_AUTO     (default) the counted scan in the launches where it is measured to pay (docs/HISTORY.md 3.1a).
 * Rayleigh targets always step; the trial-per-lane kernel (BH_KERNEL_LEAN) steps for both wave types, sixteen steps a round. */
#define BH_SCAN_STEPS 0
#define BH_SCAN_COUNTED 1
#define BH_SCAN_AUTO 2
int bh_engine_set_swd_scan(bh_engine *e, int scan);
int bh_engine_get_swd_scan(const bh_engine *e);

/* Experiment switches (csrc/bh_tuning.h lists them: name, environment variable, default, meaning).  They change scheduling
 * and launch geometry, never a result.  The table is filled once per process from the environment; this call changes one
 * entry for the calls that follow (process-wide).  BH_EINVAL for an unknown name, and always in a build with
 * -DBH_NO_EXPERIMENTS.  Not needed by a caller that only wants results. */
int bh_engine_set_tuning(bh_engine *e, const char *name, int value);
int bh_engine_get_tuning(bh_engine *e, const char *name, int *value);

/* ---- diagnostics -------------------------------------------------------------------------
 * Evaluate one elementary function on the device for n float64 inputs (host pointers):
 * op 0 sqrt, 1 sin, 2 cos, 3 exp, 4 log, 5 1/x; op 6 / 7: in holds n pairs (a, b), out[i] = a/b
 * through the shared-reciprocal sequence of the kernels (6) or the plain operator (7); op 8 / 9 / 10:
 * sin / cos / exp through the kernels' glibc-exact restatement (csrc/bh_libm.h).  Used by the tests to document how far the
 * device math library is from the host's libm (SURVEY.md 7 "FMA contraction & device libm"). */
int bh_probe_math(bh_engine *e, int op, int n, const double *in, double *out);

/* Instrumentation (off by default; bench.py and the tests turn it on).
 *   timing:   HIP events are recorded on the stream the kernels are launched on, around each
 *             kernel family of every *_batch call made after bh_timing_reset().  Nothing is
 *             synchronised until bh_timing_collect(), which waits for the events and returns the
 *             number of calls, the SUM over those calls of the span first-kernel-start ->
 *             last-kernel-end (total_ms) and of the time inside each kernel family:
 *             family_ms[0] dispersion (swd), [1] receiver function, [2] likelihood.
 *   counting: the dispersion kernels add up their secular-function evaluations; bh_last_neval()
 *             returns the count of the most recent call (the flop model of SURVEY.md 8(d) is
 *             layer-propagator steps = evaluations x (nlay-1)). */
int bh_engine_set_instrumentation(bh_engine *e, int timing, int counting);
int bh_timing_reset(bh_engine *e);
int bh_timing_collect(bh_engine *e, int *ncalls, double *total_ms, double family_ms[3]);
/* Per-call timing of the calls since bh_timing_reset(), from the same events: step_ms[i] = start of call i -> start of
 * call i+1 (back-to-back calls: one "step" each, gaps included), for the last call its own span.  Writes at most
 * `max` entries and returns their number in *n.  (No further events: a marker recorded on the launch stream between
 * two calls changes how the next call's two lane-kernel launches pair up on the SIMDs -- 8.1 -> 10.6 ms at B = 16 384.) */
int bh_timing_steps(bh_engine *e, int max, double *step_ms, int *n);
int bh_last_neval(bh_engine *e, uint64_t *neval);
/* raw counter block of the last counted call: [0] secular evaluations, [1..3] / [4..6] wave-cycles per phase
 * (development aid), [7] wavefronts, [8] / [9] evaluations of the Rayleigh / Love targets, [10] / [11] their
 * layer-propagator steps (evaluations x finite layers: the flop model of the roofline block in bench.py) */
int bh_debug_counters(bh_engine *e, uint64_t out[16]);
/* development aid: one record of 4 words per traced wavefront of the last counted dispersion launch
 * (start, end [100 MHz ticks], core cycles, rounds | wave type << 32 | HW_ID << 36); counter [7] = wavefronts */
int bh_debug_trace(bh_engine *e, uint64_t *out, int nwaves);


#ifdef __cplusplus
}
#endif
#endif
