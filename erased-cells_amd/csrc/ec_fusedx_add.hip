// Mixed-type fused expression kernels whose outer op is Add (see ec_fused_mixed_tu.hpp).
#define EC_TU_OP EC_ADD
#include "ec_fused_mixed_tu.hpp"
