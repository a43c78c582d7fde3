// bayhunter_amd/csrc/swd_group_prek.hip -- the builds of swd_group_kernel with the certified-sign scan (PREK; opt-in:
// bh_engine_set_swd_prescan), in a translation unit of their own: the same source (swd_group_kernel.hip, included below) with ONE
// wavefront per SIMD as its register budget -- the out-of-line certified evaluation does not fit beside the round loop's state
// in 256 registers (at two wavefronts per SIMD these builds carried 176-256 bytes of scratch per lane).  Only
// bh_launch_swd_group_prek is defined here.
#define BH_GROUP_PREK_TU 1
#define BH_GROUP_WAVES 1
#include "swd_group_kernel.hip"
