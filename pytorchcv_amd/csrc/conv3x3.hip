// conv3x3.hip - instantiations of the dedicated dense 3x3 / stride 1 / pad 1 kernel for the three storage types.
#include "conv3x3_inst.hpp"
CONV3_INSTANCES(CONV3_DEFINE, PCV_BF16)
CONV3_INSTANCES(CONV3_DEFINE, PCV_F16)
CONV3_INSTANCES(CONV3_DEFINE, PCV_F32)
