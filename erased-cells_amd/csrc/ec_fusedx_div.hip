// Mixed-type fused expression kernels whose outer op is Div (see ec_fused_mixed_tu.hpp).
#define EC_TU_OP EC_DIV
#include "ec_fused_mixed_tu.hpp"
