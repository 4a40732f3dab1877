// msm_g2.hip -- G2 instantiation of the Pippenger MSM (msm_impl.hpp).  Built with ZK_MUL_OUTLINE: the
// Fq2 curve formulas inline 29 base-field multiplications per mixed addition, which overflows the
// instruction cache and the register file; one out-of-line Fq multiply per TU is ~35 % faster here
// (tools/ubench_field.hip).
#define ZK_MUL_OUTLINE 1
#define ZK_MSM_FIELD Fq2
#include "msm_impl.hpp"
