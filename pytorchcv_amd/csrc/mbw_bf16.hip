// mbw_bf16.hip - bf16 instantiations of the wave-private fused inverted-residual kernel
#include "mbw_inst.hpp"
MBW_SHAPES(MBW_DEFINE, PCV_BF16)
MBW2_SHAPES(MBW2_DEFINE, PCV_BF16)
MBW3_SHAPES(MBW3_DEFINE, PCV_BF16)
