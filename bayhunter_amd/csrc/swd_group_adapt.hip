// bayhunter_amd/csrc/swd_group_adapt.hip -- the one-model-per-wavefront builds of swd_group_kernel (ADAPT: a sampler's windows, single
// models, the re-run of guarded models) in a translation unit of their own: the same source (swd_group_kernel.hip, included
// below) and the same flags; it only shortens the build (a third of that file's instantiations).  Only bh_launch_swd_group_adapt
// is defined here.
#define BH_GROUP_ADAPT_TU 1
#include "swd_group_kernel.hip"
