// ec_expr_tu.hpp — the 85 k_expr kernels whose first stream has load class EC_TU_C0 (1, 2, 4 or 8 bytes): the later
// streams' classes are packed (a class 0 = "no such stream" only after the last stream): 1 + 4 + 16 + 64 combinations.
// ec_expr_c{1,2,4,8}.hip instantiate one set each so that the four compile in parallel.
#pragma once

#include <hip/hip_runtime.h>

#include "ec_expr.hpp"

namespace ecd {

template <int C0, int C1, int C2, int C3>
constexpr ExprKernel expr_pick() {
    if constexpr ((C1 == 0 && (C2 != 0 || C3 != 0)) || (C2 == 0 && C3 != 0)) return nullptr;  // not packed: never launched
    else return k_expr<C0, C1, C2, C3>;
}

template <int C0>
ExprKernel expr_kernel(int i1, int i2, int i3) {
    if (i1 < 0 || i1 >= kFusedClasses || i2 < 0 || i2 >= kFusedClasses || i3 < 0 || i3 >= kFusedClasses) return nullptr;
#define EC_W(I1, I2, I3) case (I1 * 25 + I2 * 5 + I3): return expr_pick<C0, fused_class_bytes(I1), fused_class_bytes(I2), fused_class_bytes(I3)>();
#define EC_Z(I1, I2) EC_W(I1, I2, 0) EC_W(I1, I2, 1) EC_W(I1, I2, 2) EC_W(I1, I2, 3) EC_W(I1, I2, 4)
#define EC_Y(I1) EC_Z(I1, 0) EC_Z(I1, 1) EC_Z(I1, 2) EC_Z(I1, 3) EC_Z(I1, 4)
    switch (i1 * 25 + i2 * 5 + i3) {
        EC_Y(0) EC_Y(1) EC_Y(2) EC_Y(3) EC_Y(4)
    }
#undef EC_Y
#undef EC_Z
#undef EC_W
    return nullptr;
}

}  // namespace ecd

#ifdef EC_TU_C0
namespace ecd {
template ExprKernel expr_kernel<EC_TU_C0>(int, int, int);
}
#endif
