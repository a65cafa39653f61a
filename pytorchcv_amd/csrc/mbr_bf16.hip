// mbr_bf16.hip - bf16 instantiations of the register-resident fused inverted-residual kernel
#include "mbr_inst.hpp"
MBR_SHAPES(MBR_DEFINE, PCV_BF16)
