// hconv3x3.hip - explicit instantiations of the halo-reuse 3x3 kernel (one translation unit keeps pcv_api.hip's build short)
#include <hip/hip_runtime.h>
#include "hconv3x3_inst.hpp"
HCONV_INSTANCES(HCONV_DEFINE)
