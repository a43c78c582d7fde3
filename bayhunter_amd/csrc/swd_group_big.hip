// bayhunter_amd/csrc/swd_group_big.hip -- the one build of swd_group_kernel that needs more than 256 registers (one model per
// wavefront, both sequences, the counted Love scan AND the counters and clocks: an instrumented launch of a sampler's window under
// BH_SEARCH_FAST_RAYLEIGH), in a translation unit of its own: the same source (swd_group_kernel.hip, included below) with ONE
// wavefront per SIMD as its register budget.  Only bh_launch_swd_group_big is defined here.
#define BH_GROUP_BIG_TU 1
#define BH_GROUP_WAVES 1
#include "swd_group_kernel.hip"
