// kern_unit.hip -- one translation unit of heavy template-kernel instantiations.  Compiled once per group with
// -DGKR_GROUP_<NAME> (gkr-mimc_amd/build.py: UNITS; the groups are listed in kernel_groups.h); gkrhip.hip declares the same
// instantiations `extern template` and launches them.  No host code here beyond the kernels' launch stubs.
#define GKR_KERNEL_TU
#include <hip/hip_runtime.h>
#if defined(GKR_GROUP_MSM_G1) || defined(GKR_GROUP_MSM_G2A) || defined(GKR_GROUP_MSM_G2B) || defined(GKR_GROUP_MSM_G2C)
#include "g1.hip.h"
#elif defined(GKR_GROUP_NTT)
#include "ntt.hip.h"
#elif defined(GKR_GROUP_WIDE2) || defined(GKR_GROUP_WIDEPRE)
#include "cipher_round.hip.h"
#elif defined(GKR_GROUP_ROUND)
#include "cipher_round.hip.h"
#include "linear_round.hip.h"
#include "cipher_coop.hip.h"
#else
#error "kern_unit.hip: no GKR_GROUP_<NAME> given"
#endif
#define GKR_INST
#include "kernel_groups.h"
