// d3q_bf16.hip - bf16 instantiations of the 8-wave dense 3x3 kernel with filter-row reuse
#include "d3q_inst.hpp"
D3Q_SHAPES(D3Q_DEFINE, PCV_BF16)
D3Q1_SHAPES(D3Q1_DEFINE, PCV_BF16)
