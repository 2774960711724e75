// Expression-program kernels whose first stream has load class 8 (see ec_expr_tu.hpp).
#define EC_TU_C0 8
#include "ec_expr_tu.hpp"
