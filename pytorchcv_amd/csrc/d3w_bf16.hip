// d3w_bf16.hip - bf16 instantiations of the large-tile dense 3x3 kernel (eight self-loading waves)
#include "d3w_inst.hpp"
D3W_SHAPES(D3W_DEFINE, PCV_BF16)
D3WT_SHAPES(D3WT_DEFINE, PCV_BF16)
