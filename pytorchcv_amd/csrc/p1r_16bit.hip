// p1r_16bit.hip - bf16 and fp16 instantiations of the 1x1 kernel with register-resident weights (256 / 512 input channels)
#include "p1r_conv.hpp"
#define P1R_INST(CW, CIN)                                                   \
    template __global__ void p1r_kernel<PCV_BF16, CW, CIN>(const D3Params); \
    template __global__ void p1r_kernel<PCV_F16, CW, CIN>(const D3Params);
P1R_INST(64, 256)
P1R_INST(32, 512)
P1R_INST(32, 256)
