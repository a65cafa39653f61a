// d1i_kernel (d1i_conv.hpp): both 16-bit types, 1024 / 2048 input channels
#include "d1i_conv.hpp"
template __global__ void d1i_kernel<PCV_BF16, 1024>(const D3Params);
template __global__ void d1i_kernel<PCV_F16, 1024>(const D3Params);
template __global__ void d1i_kernel<PCV_BF16, 2048>(const D3Params);
template __global__ void d1i_kernel<PCV_F16, 2048>(const D3Params);
