// Mixed-type fused expression kernels whose outer op is Sub (see ec_fused_mixed_tu.hpp).
#define EC_TU_OP EC_SUB
#include "ec_fused_mixed_tu.hpp"
