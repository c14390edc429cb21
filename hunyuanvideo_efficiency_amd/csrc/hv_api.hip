#include "hv_common.hpp"
#include "../../include/hv_kernels.h"
extern "C" int hv_abi_version(void) { return 1; }
