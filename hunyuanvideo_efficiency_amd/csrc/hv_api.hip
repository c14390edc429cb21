#include "hv_common.hpp"
#include "../../include/hv_kernels.h"
extern "C" int hv_abi_version(void) { return 4; }   // 2: + hv_euler_step_f32_f32, hv_gemm_fp8 family; 3: conv gn_partial, sub-pixel upsampler conv; 4: hv_groupnorm_finalize_f16 takes partial_floats
