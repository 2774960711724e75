// ec_fused_mixed_tu.hpp — launchers of the mixed-type fused kernels for one outer op O2:
// 12 ordered type pairs x (A B A B: 4 inner ops x 4 second-term ops  +  3 three-operand patterns x 4 inner ops)
// = 336 kernels; ec_fusedx_{add,sub,mul,div}.hip instantiate one each.
#pragma once

#include <hip/hip_runtime.h>

#include "ec_fused_mixed.hpp"
#include "ec_runtime.hpp"

namespace ecd {

template <typename A, typename B, int O1, int O2>
static bool launch_mixed_pat(const FusedArgs& fa, int pat, unsigned grid, double* out, uint8_t* om, size_t n, hipStream_t s) {
    if (fa.o3 == kOpNone) {
        switch (pat) {
            case kPatAAB: k_fused_mixed<A, B, kPatAAB, O1, O2, kOpNone><<<grid, kBlock, 0, s>>>(fa, out, om, n); return true;
            case kPatABA: k_fused_mixed<A, B, kPatABA, O1, O2, kOpNone><<<grid, kBlock, 0, s>>>(fa, out, om, n); return true;
            case kPatABB: k_fused_mixed<A, B, kPatABB, O1, O2, kOpNone><<<grid, kBlock, 0, s>>>(fa, out, om, n); return true;
            default: return false;
        }
    }
    if (pat != kPatABAB) return false;
    switch (fa.o3) {
        case EC_ADD: k_fused_mixed<A, B, kPatABAB, O1, O2, EC_ADD><<<grid, kBlock, 0, s>>>(fa, out, om, n); return true;
        case EC_SUB: k_fused_mixed<A, B, kPatABAB, O1, O2, EC_SUB><<<grid, kBlock, 0, s>>>(fa, out, om, n); return true;
        case EC_MUL: k_fused_mixed<A, B, kPatABAB, O1, O2, EC_MUL><<<grid, kBlock, 0, s>>>(fa, out, om, n); return true;
        case EC_DIV: k_fused_mixed<A, B, kPatABAB, O1, O2, EC_DIV><<<grid, kBlock, 0, s>>>(fa, out, om, n); return true;
        default: return false;
    }
}

template <typename A, typename B, int O2>
static bool launch_mixed_o1(const FusedArgs& fa, int pat, unsigned grid, double* out, uint8_t* om, size_t n, hipStream_t s) {
    switch (fa.o1) {
        case EC_ADD: return launch_mixed_pat<A, B, EC_ADD, O2>(fa, pat, grid, out, om, n, s);
        case EC_SUB: return launch_mixed_pat<A, B, EC_SUB, O2>(fa, pat, grid, out, om, n, s);
        case EC_MUL: return launch_mixed_pat<A, B, EC_MUL, O2>(fa, pat, grid, out, om, n, s);
        default: return launch_mixed_pat<A, B, EC_DIV, O2>(fa, pat, grid, out, om, n, s);
    }
}

// false: no kernel for this (pair, pattern) — the caller falls back to convert-then-fuse
template <int O2>
bool dispatch_fused_mixed(const FusedArgs& fa, int pair, int pat, unsigned grid, double* out, uint8_t* om, size_t n, hipStream_t s) {
    switch (pair) {
#define EC_ROW(IDX, AID, AT, BID, BT) case IDX: return launch_mixed_o1<AT, BT, O2>(fa, pat, grid, out, om, n, s);
        EC_FUSED_MIXED_PAIRS(EC_ROW)
#undef EC_ROW
        default: return false;
    }
}

}  // namespace ecd

#ifdef EC_TU_OP
namespace ecd {
template bool dispatch_fused_mixed<EC_TU_OP>(const FusedArgs&, int, int, unsigned, double*, uint8_t*, size_t, hipStream_t);
}
#endif
