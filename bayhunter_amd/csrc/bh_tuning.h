// bayhunter_amd/csrc/bh_tuning.h -- the library's experiment switches, in ONE place.
//
// Everything here changes scheduling, launch geometry or which of two equivalent code paths runs -- never a result that is
// part of the contract of include/bh_engine.h.  The table is filled ONCE per process, from the environment, the first time it
// is looked at (bh_engine_create does); nothing reads the environment per call.  Tests and tools that need to flip a switch
// inside a process use bh_engine_set_tuning(name, value) (the name is the field's name below).  A build with
// -DBH_NO_EXPERIMENTS ignores the environment and refuses bh_engine_set_tuning: every switch has its default.
//
// The ENGINE SETTINGS with an API of their own (bh_engine_set_swd_search / _scan / _arith / _group / _lookahead) take their
// initial value from the first five entries; an API call afterwards wins.
#pragma once

//        field              environment variable        default  meaning (value: flag = set to anything / integer)
#define BH_TUNING_TABLE(X)                                                                                                          \
    X(swd_search,        "BH_SWD_SEARCH",        -1, "initial root refinement: r(eference) / f(ast) / fast_rayleigh; -1 = the library's default")    \
    X(swd_scan,          "BH_SWD_SCAN",          -1, "initial scan mode: s(teps) / c(ounted) / a(uto); -1 = auto")                                  \
    X(swd_arith,         "BH_SWD_ARITH",         -1, "initial arithmetic of short-refinement launches: e(xact) / f(ast); -1 = the library's default") \
    X(swd_group,         "BH_SWD_GROUP",          0, "lanes per model of the dispersion kernel (0 = planned per launch)")                           \
    X(swd_lookahead,     "BH_SWD_LOOKAHEAD",      0, "trial velocities per round (0 = planned per launch)")                                         \
    X(swd_look_r,        "BH_SWD_LOOK_R",         0, "trials per round of Rayleigh wavefronts only (0 = planned)")                                  \
    X(swd_look_l,        "BH_SWD_LOOK_L",         0, "trials per round of Love wavefronts only (0 = planned)")                                      \
    X(swd_love_inlook,   "BH_SWD_LOVE_INLOOK",    0, "Love trials inside a lane group, 1..4 (0 = automatic)")                                       \
    X(swd_prio_low,      "BH_SWD_PRIO_LOW",      -1, "issue priority of a dispersion wavefront's unfavoured phase beside RF wavefronts (-1 = 1)")    \
    X(swd_pair_minwaves, "BH_SWD_PAIR_MINWAVES", -1, "wavefronts from which the SIMD-pairing order of the models is used (-1 = 7 x CUs)")           \
    X(swd_slice,         "BH_SWD_SLICE",          0, "lane-per-model kernel: rounds between priority changes (0 = default)")                        \
    X(swd_hint_always,   "BH_SWD_HINT_ALWAYS",    0, "flag: honour the typical-depth hint also when every model gets a wavefront")                   \
    X(swd_no_pair,       "BH_SWD_NO_PAIR",        0, "flag: no SIMD-pairing order of the models")                                                    \
    X(swd_no_mix,        "BH_SWD_NO_MIX",         0, "flag: Rayleigh and Love wavefronts not interleaved in the grid")                               \
    X(swd_no_adapt,      "BH_SWD_NO_ADAPT",       0, "flag: one-model-per-wavefront launches do not size lane groups per model")                     \
    X(swd_no_restart,    "BH_SWD_NO_RESTART",     0, "flag: guarded models are re-run by a second launch instead of restarting in place")            \
    X(swd_no_simple,     "BH_SWD_NO_SIMPLE",      0, "flag: no kernel build specialised for fundamental-mode phase velocities")                      \
    X(swd_no_board,      "BH_SWD_NO_BOARD",       0, "flag: no progress board between the two wavefronts of a SIMD")                                 \
    X(swd_no_fair,       "BH_SWD_NO_FAIR",        0, "flag: no alternating issue priority of the two wavefronts of a SIMD")                          \
    X(swd_redundant,     "BH_SWD_REDUNDANT",      0, "flag: every lane of a group runs the whole Rayleigh recursion")                                \
    X(swd_no_lean,       "BH_SWD_NO_LEAN",        0, "flag: fast-arithmetic launches take the group / lane kernels' FA builds, not the trial-per-lane kernel") \
    X(swd_lean_pairs,    "BH_SWD_LEAN_PAIRS",     0, "trial-per-lane kernel: largest call in (model, target) pairs that takes it (0 = 2^20)")        \
    X(swd_lean_no_sort,  "BH_SWD_LEAN_NO_SORT",   0, "flag: trial-per-lane kernel: models by depth only, not by predicted search length")           \
    X(swd_lean_r,        "BH_SWD_LEAN_R",         0, "trial-per-lane kernel: trials per round of Rayleigh targets, power of two 4..64 (0 = planned)")  \
    X(swd_lean_l,        "BH_SWD_LEAN_L",         0, "trial-per-lane kernel: trials per round of Love targets (0 = planned)")                         \
    X(swd_lean_xcd,      "BH_SWD_LEAN_XCD",       1, "trial-per-lane kernel: 1 = the models ordered inside eight blocks of the batch, a block per XCD (0: one order for the chip)") \
    X(swd_rerun_wgs,     "BH_SWD_RERUN_WGS",    256, "workgroups per target of the re-run launch of guarded models, striding over the list (0 = one wavefront per model of the batch, as before round 6)") \
    X(swd_gsplit,        "BH_SWD_GSPLIT",  16777216, "group velocities (fundamental mode): calls of up to this many (model, period) pairs per target -- 2^24, the most the launch indexes -- run the chain of first roots and the second roots as two launches (0 = one search after the other in one launch)") \
    X(swd_lean_flip,     "BH_SWD_LEAN_FLIP",    256, "trial-per-lane kernel, two targets: 256 = the second target takes the models in the opposite order (0: the same order); + n: the targets of a wavefront pair swap with bit n-1 of the workgroup index") \
    X(no_order,          "BH_NO_ORDER",           0, "flag: models processed in the caller's order")                                                 \
    X(no_overlap,        "BH_NO_OVERLAP",         0, "flag: receiver function after the dispersion kernel, not beside it")                           \
    X(no_started,        "BH_NO_STARTED",         0, "flag: no start gate (stream memory operation) for the receiver-function stream")               \
    X(no_mfma,           "BH_NO_MFMA",            0, "flag: Gauss-law quadratic form in like_kernel instead of the MFMA contraction")                \
    X(err_memset,        "BH_ERR_MEMSET",         0, "flag: zero the per-target failure flags on every call")                                        \
    X(gauss_tile,        "BH_GAUSS_TILE",         0, "tile of the Gauss-law contraction: 64 / 128 (0 = automatic)")                                  \
    X(rf_lds_beside,     "BH_RF_LDS_BESIDE",     -1, "LDS floor (bytes) of receiver-function workgroups beside an ungated dispersion launch (-1 = default)") \
    X(rf_lds_gated,      "BH_RF_LDS_GATED",       0, "LDS floor (bytes) of receiver-function workgroups of a gated fused call (0 = none)")           \
    X(rf_keep_floor,     "BH_RF_KEEP_FLOOR",      0, "flag: keep the ungated LDS floor in a gated fused call")                                       \
    X(rf_coef_big,       "BH_RF_COEF_BIG",        0, "flag: the large-register coefficient kernel also beside the dispersion kernel")                \
    X(rf_no_fuse,        "BH_RF_NO_FUSE",         0, "flag: the receiver function always writes its trace (no fused likelihood)")                    \
    X(rf_no_cut,         "BH_RF_NO_CUT",          0, "flag: every frequency bin computed (no spectral cut-off below 1e-17)")                         \
    X(rf_no_realc,       "BH_RF_NO_REALC",        0, "flag: always the general (complex-coefficient) recursion")                                     \
    X(rf_no_rot,         "BH_RF_NO_ROT",          0, "flag: no rotation of the bins over a workgroup's wavefronts")                                  \
    X(rf_threads,        "BH_RF_THREADS",         0, "threads of a synthesis workgroup: 128 (0 = 256)")                                              \
    X(rf_waves,          "BH_RF_WAVES",           0, "register budget of the synthesis kernel: 3 wavefronts per SIMD (0 = 4)")                       \
    X(debug_plan,        "BH_DEBUG_PLAN",         0, "flag: print launch plans to stderr")

struct BhTuning {
#define X(field, env, dflt, doc) int field = dflt;
    BH_TUNING_TABLE(X)
#undef X
    int under_pmc = 0; // ROCPROF_COUNTER_COLLECTION is set: rocprofv3 --pmc serialises dispatches, so the start gate is off
};

const BhTuning &bh_tuning();                      // parsed on first use
int bh_tuning_set(const char *name, int value);   // 0 ok, -1 unknown name or BH_NO_EXPERIMENTS
int bh_tuning_get(const char *name, int *value);  // 0 ok, -1 unknown name
