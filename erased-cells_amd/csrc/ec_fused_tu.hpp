// ec_fused_tu.hpp — launchers of the fused kernels for one outer op O2 (4 inner ops x 5 second-term
// ops x 10 cell types); ec_fused_{add,sub,mul,div}.hip instantiate one each.
#pragma once

#include <hip/hip_runtime.h>

#include "ec_fused_kernels.hpp"
#include "ec_runtime.hpp"

namespace ecd {

template <int O1, int O2, int O3>
static void launch_fused_ops(const FusedArgs& fa, int same_dt, unsigned grid, double* out, uint8_t* out_mask, size_t n, hipStream_t s) {
    switch (same_dt) {
#define EC_ROW(ID, T) case ID: k_fused_same<T, O1, O2, O3><<<grid, kBlock, 0, s>>>(fa, out, out_mask, n); return;
        EC_WITH_CT(EC_ROW)
#undef EC_ROW
        default: return;  // unreachable: the caller unifies mixed operand types first
    }
}

template <int O1, int O2>
static void launch_fused_o3(const FusedArgs& fa, int same_dt, unsigned grid, double* out, uint8_t* out_mask, size_t n, hipStream_t s) {
    switch (fa.o3) {
        case EC_ADD: return launch_fused_ops<O1, O2, EC_ADD>(fa, same_dt, grid, out, out_mask, n, s);
        case EC_SUB: return launch_fused_ops<O1, O2, EC_SUB>(fa, same_dt, grid, out, out_mask, n, s);
        case EC_MUL: return launch_fused_ops<O1, O2, EC_MUL>(fa, same_dt, grid, out, out_mask, n, s);
        case EC_DIV: return launch_fused_ops<O1, O2, EC_DIV>(fa, same_dt, grid, out, out_mask, n, s);
        default: return launch_fused_ops<O1, O2, kOpNone>(fa, same_dt, grid, out, out_mask, n, s);
    }
}

// same_dt: the common cell type of all buffer operands
template <int O2>
void dispatch_fused(const FusedArgs& fa, int same_dt, unsigned grid, double* out, uint8_t* out_mask, size_t n, hipStream_t s) {
    switch (fa.o1) {
        case EC_ADD: return launch_fused_o3<EC_ADD, O2>(fa, same_dt, grid, out, out_mask, n, s);
        case EC_SUB: return launch_fused_o3<EC_SUB, O2>(fa, same_dt, grid, out, out_mask, n, s);
        case EC_MUL: return launch_fused_o3<EC_MUL, O2>(fa, same_dt, grid, out, out_mask, n, s);
        default: return launch_fused_o3<EC_DIV, O2>(fa, same_dt, grid, out, out_mask, n, s);
    }
}

}  // namespace ecd

#ifdef EC_TU_OP
namespace ecd {
template void dispatch_fused<EC_TU_OP>(const FusedArgs&, int, unsigned, double*, uint8_t*, size_t, hipStream_t);
}
#endif
