// d3i_kernel (d3i_conv.hpp): both 16-bit types
#include "d3i_conv.hpp"
template __global__ void d3i_kernel<PCV_BF16>(const D3Params);
template __global__ void d3i_kernel<PCV_F16>(const D3Params);
