// ec_fused_kernels.hpp — what the fused two-level expression kernels share:  out = (x o1 y) o2 (z o3 w)  in one pass.
//
// the reference evaluates operator chains eagerly — `(&nir - &red) / (nir + red)`
// (src/gdal/rasterband.rs:148,178) is three passes with two f64 temporaries, `(buf + ones) * 2.0`
// (examples/masked.rs:12) two passes.  Every intermediate of the eager chain is already an f64
// rounded once per step (src/value.rs:207), so evaluating the same steps in registers gives
// bit-identical results while the temporaries never touch HBM:
//   NDVI on u16:      eager 12 + 12 + 24 = 48 B/cell  ->  fused 2 + 2 + 8 = 12 B/cell
//   (a+b)*c f32+mask: eager 19 + 23 = 42 B/cell       ->  fused 4+4+4 + 8 + 3+1 = 24 B/cell
//
// Here: the argument block, the single-cell evaluation (peeled head / odd tail cell, and the cell-wise comparison
// kernel), the mask phase.  The streaming kernel itself is k_fused_any (ec_fused_any.hpp): one family for every mix of
// operand cell types.  (Rounds 1-2 shipped kernels specialised per cell type and op triple — k_fused_same, 800
// instantiations — and per ordered type pair — k_fused_mixed, 1,344; measured against k_fused_any on the same buffers
// they are slower or equal on every workload (profiles/r03/tune_fused_any_u2.log: NDVI u16 0.73 vs 0.81 of peak, NDVI
// u16 + f32 0.77 vs 0.81, config 3 0.80 vs 0.79) and were removed; tools/legacy_fused_kernels.hpp keeps them for the A/B.)
#pragma once

#include "ec_binop_kernels.hpp"

namespace ecd {

using D2 = vec<double, 2>;

constexpr int kOpNone = -1;  // o3 == kOpNone: the second term is z alone (three-operand chain)

struct FusedArgs {
    const void* p[4];        // x, y, z, w (device)
    const uint8_t* m[4];     // masks or null
    int8_t dt[4];            // cell types
    int8_t alias[4];         // alias[k] = j < k if operand k is the same buffer as operand j, else k
    int8_t o1, o2, o3;
    int8_t nmask;            // number of distinct masks to AND (0 = unmasked call)
    int8_t is_sc[4];         // operand k is a scalar constant (no stream): value sc[k]
    uint8_t head;            // leading cells (0/1) computed singly so the pair loads of 1-byte cells start on even
                             // addresses (peel_head, ec_runtime.hpp); vector kernel only
    uint8_t small;           // NDVI shape over ≤16-bit integer buffers: the short exact divide (ec_fused_any.hpp)
    uint8_t cacheable;       // load policy of the launch (cache_plan, ec_runtime.hpp): bit k = operand slot k's own stream,
                             // bit 4 + j = mask m[j] is loaded with the default cache policy instead of nt
    double sc[4];
};

template <typename T>
__device__ __forceinline__ double load_cell_as(const void* p, size_t i) { return to_f64(ld_cell(static_cast<const T*>(p) + i)); }

__device__ __forceinline__ double load_cell_f64(const void* p, int dt, size_t i) {
    switch (dt) {
#define EC_ROW(ID, T) case ID: return load_cell_as<T>(p, i);
        EC_WITH_CT(EC_ROW)
#undef EC_ROW
    }
    return 0.0;
}

__device__ __forceinline__ double operand_cell(const FusedArgs& fa, int k, size_t i) {
    return fa.is_sc[k] ? fa.sc[k] : load_cell_f64(fa.p[k], fa.dt[k], i);
}

__device__ __forceinline__ double apply_rt(int op, double a, double b) {
    switch (op) {  // wave-uniform
        case EC_ADD: return cell_op<EC_ADD, true>(a, b);
        case EC_SUB: return cell_op<EC_SUB, true>(a, b);
        case EC_MUL: return cell_op<EC_MUL, true>(a, b);
        default: return cell_op<EC_DIV, true>(a, b);
    }
}

// run-time ops (cell-wise fallback kernel only)
__device__ __forceinline__ double fused_cell(const FusedArgs& fa, double x, double y, double z, double w) {
    const double t1 = apply_rt(fa.o1, x, y);
    const double t2 = fa.o3 == kOpNone ? z : apply_rt(fa.o3, z, w);
    return apply_rt(fa.o2, t1, t2);
}

// Pairs per lane per tile of k_fused_any.  2 for every mix of operand widths (tools/tune_fused_any.hip built with
// -DEC_FUSED_U=1 / 2 / by-narrowest-stream 4: profiles/r03/tune_fused_any_u*.log — NDVI u16 0.71 / 0.81 / 0.81 of peak,
// NDVI u16 + f32 0.71 / 0.81 / 0.68, (u16*u16)+(f32*f32) 0.73 / 0.81 / 0.64, config 3 0.83 / 0.79 / 0.76).
#ifdef EC_FUSED_U
constexpr int fused_u(size_t) { return EC_FUSED_U; }
#else
constexpr int fused_u(size_t) { return 2; }
#endif

// mask phase: AND of the distinct operand masks (src/masked/masked_buffer.rs:333 applied per step),
// 16 mask bytes per lane
__device__ __forceinline__ void fused_mask_phase(const FusedArgs& fa, uint8_t* __restrict__ out_mask, size_t n) {
    if (fa.nmask > 0) {
        const size_t ngroups = n / 16;
        const size_t stride = size_t(gridDim.x) * kBlock;
        u32x4* __restrict__ om = reinterpret_cast<u32x4*>(out_mask);
        for (size_t g = size_t(blockIdx.x) * kBlock + threadIdx.x; g < ngroups; g += stride) {
            u32x4 acc = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            for (int k = 0; k < fa.nmask; ++k) {
                const u32x4* mk = reinterpret_cast<const u32x4*>(fa.m[k]) + g;
                u32x4 x;
                policy_arms<1>(fa.cacheable >> (4 + k), [&](auto bits) { x = load_vec<!(decltype(bits)::value & 1u)>(mk); });  // launch-uniform
                acc &= x;
            }
            mask_store(acc, om + g);
        }
        if (blockIdx.x == 0)
            for (size_t i = ngroups * 16 + threadIdx.x; i < n; i += kBlock) {
                uint8_t acc = ld_cell(fa.m[0] + i);
                for (int k = 1; k < fa.nmask; ++k) acc &= ld_cell(fa.m[k] + i);
                st_cell(acc, out_mask + i);
            }
    }
}


// Any alignment: one cell per lane, run-time ops.  (A template only so that the header can be included
// by several translation units.)
template <int UNUSED = 0>
__global__ __launch_bounds__(kBlock) void k_fused_cellwise(FusedArgs fa, double* __restrict__ out, uint8_t* __restrict__ out_mask, size_t n) {
    const size_t stride = size_t(gridDim.x) * kBlock;
    const bool has_w = fa.o3 != kOpNone;
    for (size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        out[i] = fused_cell(fa, operand_cell(fa, 0, i), operand_cell(fa, 1, i), operand_cell(fa, 2, i),
                            has_w ? operand_cell(fa, 3, i) : 0.0);
        if (fa.nmask > 0) {
            uint8_t acc = fa.m[0][i];
            for (int k = 1; k < fa.nmask; ++k) acc &= fa.m[k][i];
            out_mask[i] = acc;
        }
    }
}

}  // namespace ecd
