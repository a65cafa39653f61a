// d3c_16bit.hip - bf16 and fp16 instantiations of the 64-input-channel dense 3x3 kernel (weights in registers)
#include "d3c_conv.hpp"
template __global__ void d3c_kernel<PCV_BF16>(const D3Params);
template __global__ void d3c_kernel<PCV_F16>(const D3Params);
