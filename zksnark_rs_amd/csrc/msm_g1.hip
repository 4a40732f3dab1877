// msm_g1.hip -- G1 instantiation of the Pippenger MSM (msm_impl.hpp) + the shared sorting kernels.
// no scheduling fences here: at 131 VGPRs (3 waves/SIMD) letting the scheduler interleave independent
// multiplications of the G1 formulas is 5 % faster than 121 VGPRs / 4 waves with fences (bench r1)
#define ZK_NO_SCHED_FENCE 1
#define ZK_MSM_COMMON 1
#define ZK_MSM_FIELD Fq
#include "msm_impl.hpp"
