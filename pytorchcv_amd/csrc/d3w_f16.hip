// d3w_f16.hip - fp16 instantiations of the large-tile dense 3x3 kernel (eight self-loading waves)
#include "d3w_inst.hpp"
D3W_SHAPES(D3W_DEFINE, PCV_F16)
D3WT_SHAPES(D3WT_DEFINE, PCV_F16)
