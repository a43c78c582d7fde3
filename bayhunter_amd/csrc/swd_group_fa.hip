// bayhunter_amd/csrc/swd_group_fa.hip -- the builds of swd_group_kernel with the FAST ARITHMETIC (swd_fa.h), in a translation
// unit of their own: the same source (swd_group_kernel.hip, included below), compiled with this file's flags (Makefile) and
// its own register budget.  Only bh_launch_swd_group_fa is defined here.
#define BH_GROUP_FA_TU 1
#ifndef BH_GROUP_WAVES
#define BH_GROUP_WAVES 2
#endif
#include "swd_group_kernel.hip"
