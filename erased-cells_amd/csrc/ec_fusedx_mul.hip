// Mixed-type fused expression kernels whose outer op is Mul (see ec_fused_mixed_tu.hpp).
#define EC_TU_OP EC_MUL
#include "ec_fused_mixed_tu.hpp"
