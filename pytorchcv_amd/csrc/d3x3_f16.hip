// d3x3_f16.hip - fp16 instantiations of the 8-wave dense 3x3 kernel
#include "d3x3_inst.hpp"
D3X3_SHAPES(D3X3_DEFINE, PCV_F16)
