// Fused kernels over operands of any cell-type mix whose x slot has load class 8 (see ec_fused_any_tu.hpp).
#define EC_TU_CX 8
#include "ec_fused_any_tu.hpp"
