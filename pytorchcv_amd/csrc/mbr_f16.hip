// mbr_f16.hip - fp16 instantiations of the register-resident fused inverted-residual kernel
#include "mbr_inst.hpp"
MBR_SHAPES(MBR_DEFINE, PCV_F16)
