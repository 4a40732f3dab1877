// msm_g1.hip -- G1 instantiation of the Pippenger MSM (msm_impl.hpp) + the shared sorting kernels.
#define ZK_MSM_COMMON 1
#define ZK_MSM_FIELD Fq
#include "msm_impl.hpp"
