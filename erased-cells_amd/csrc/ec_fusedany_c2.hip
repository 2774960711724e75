// Fused kernels over operands of any cell-type mix whose x slot has load class 2 (see ec_fused_any_tu.hpp).
#define EC_TU_CX 2
#include "ec_fused_any_tu.hpp"
