// Expression-program kernels whose first stream has load class 1 (see ec_expr_tu.hpp).
#define EC_TU_C0 1
#include "ec_expr_tu.hpp"
