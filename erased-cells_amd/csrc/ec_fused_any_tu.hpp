// ec_fused_any_tu.hpp — the 125 (124 for class 0) k_fused_any kernels whose x slot has load class EC_TU_CX (0, 1, 2, 4 or 8 bytes);
// ec_fusedany_c{0,1,2,4,8}.hip instantiate one set each so that the five compile in parallel.
#pragma once

#include <hip/hip_runtime.h>

#include "ec_fused_any.hpp"

namespace ecd {

// <0, 0, 0, 0> (no operand is a buffer) is refused by the ABI and not instantiated
template <int CX, int CY, int CZ, int CW>
constexpr FusedAnyKernel fused_any_pick() {
    if constexpr (CX == 0 && CY == 0 && CZ == 0 && CW == 0) return nullptr;
    else return k_fused_any<CX, CY, CZ, CW>;
}

template <int CX>
FusedAnyKernel fused_any_kernel(int iy, int iz, int iw) {
    if (iy < 0 || iy >= kFusedClasses || iz < 0 || iz >= kFusedClasses || iw < 0 || iw >= kFusedClasses) return nullptr;
#define EC_W(IY, IZ, IW) case (IY * 25 + IZ * 5 + IW): return fused_any_pick<CX, fused_class_bytes(IY), fused_class_bytes(IZ), fused_class_bytes(IW)>();
#define EC_Z(IY, IZ) EC_W(IY, IZ, 0) EC_W(IY, IZ, 1) EC_W(IY, IZ, 2) EC_W(IY, IZ, 3) EC_W(IY, IZ, 4)
#define EC_Y(IY) EC_Z(IY, 0) EC_Z(IY, 1) EC_Z(IY, 2) EC_Z(IY, 3) EC_Z(IY, 4)
    switch (iy * 25 + iz * 5 + iw) {
        EC_Y(0) EC_Y(1) EC_Y(2) EC_Y(3) EC_Y(4)
    }
#undef EC_Y
#undef EC_Z
#undef EC_W
    return nullptr;
}

}  // namespace ecd

#ifdef EC_TU_CX
namespace ecd {
template FusedAnyKernel fused_any_kernel<EC_TU_CX>(int, int, int);
}
#endif
