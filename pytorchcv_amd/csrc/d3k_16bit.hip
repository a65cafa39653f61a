// d3k_16bit.hip - bf16 and fp16 instantiations of the 128-input-channel dense 3x3 kernel for 28-wide maps (weights in registers / AGPRs)
#include "d3k_conv.hpp"
template __global__ void d3k_kernel<PCV_BF16>(const D3Params);
template __global__ void d3k_kernel<PCV_F16>(const D3Params);
