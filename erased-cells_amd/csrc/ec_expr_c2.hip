// Expression-program kernels whose first stream has load class 2 (see ec_expr_tu.hpp).
#define EC_TU_C0 2
#include "ec_expr_tu.hpp"
