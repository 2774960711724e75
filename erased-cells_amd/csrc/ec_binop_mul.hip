// One arithmetic op × 100 operand-type pairs (see ec_binop_tu.hpp).
#define EC_TU_OP EC_MUL
#include "ec_binop_tu.hpp"
