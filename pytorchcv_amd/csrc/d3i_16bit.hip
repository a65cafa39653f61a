// d3i_kernel (d3i_conv.hpp): both 16-bit types, 256 input channels (maps up to 14 x 14) and 512 (up to 7 x 7, two images per block)
#include "d3i_conv.hpp"
template __global__ void d3i_kernel<PCV_BF16, 256>(const D3Params);
template __global__ void d3i_kernel<PCV_F16, 256>(const D3Params);
template __global__ void d3i_kernel<PCV_BF16, 512>(const D3Params);
template __global__ void d3i_kernel<PCV_F16, 512>(const D3Params);
