// Fused kernels over operands of any cell-type mix whose x slot has load class 1 (see ec_fused_any_tu.hpp).
#define EC_TU_CX 1
#include "ec_fused_any_tu.hpp"
