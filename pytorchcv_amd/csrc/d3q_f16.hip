// d3q_f16.hip - fp16 instantiations of the 8-wave dense 3x3 kernel with filter-row reuse
#include "d3q_inst.hpp"
D3Q_SHAPES(D3Q_DEFINE, PCV_F16)
D3Q1_SHAPES(D3Q1_DEFINE, PCV_F16)
