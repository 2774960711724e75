// Expression-program kernels whose first stream has load class 4 (see ec_expr_tu.hpp).
#define EC_TU_C0 4
#include "ec_expr_tu.hpp"
