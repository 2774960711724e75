// ec_fused_mixed.hpp — the fused two-level expression over buffer operands of TWO cell types, in one pass.
//
// `(&nir - &red) / (nir + red)` with a UInt16 and a Float32 band, `(a + b) * c` with one wider operand: the
// same-type kernel (ec_fused_kernels.hpp) needs every buffer operand widened to the common CellType::union first —
// a convert pass per narrower operand through pooled temporaries: 22 B/cell of traffic for 14 algorithmic on NDVI
// u16 + f32, an allocation inside the call, and no hipGraph capture.  Here each operand SLOT has a compile-time
// cell type, so every load is the typed, lane-contiguous pair load of k_fused_same and the widening to f64 happens
// in registers (what `unify` + `to_f64` of src/value.rs:103-107,207 amount to — value-preserving).
//
// A slot pattern says which of the two types (A = the first, B = the second of an ordered pair) each slot has.
// Instantiated: the four-operand pattern A B A B (NDVI: x, z of one type, y, w of the other) and the three
// three-operand patterns A A B, A B A, A B B, for the ordered pairs listed in EC_FUSED_MIXED_PAIRS — the unions
// the raster callers produce.  A scalar operand fits any slot.  Everything else falls back to
// convert-then-fuse.  (A run-time typed loader was tried in round 1: 27 % of peak — its per-cell type switch
// serialises the loads; the per-slot compile-time types are what make this a streaming kernel.)
#pragma once

#include <type_traits>

#include "ec_fused_kernels.hpp"

namespace ecd {

// ordered (A, B) pairs: both orders of (u16, f32), (u8, u16), (i16, f32), (f32, f64), (u8, f32), (u16, i16)
#define EC_FUSED_MIXED_PAIRS(X)        \
    X(0, EC_U16, uint16_t, EC_F32, float)   \
    X(1, EC_F32, float, EC_U16, uint16_t)   \
    X(2, EC_U8, uint8_t, EC_U16, uint16_t)  \
    X(3, EC_U16, uint16_t, EC_U8, uint8_t)  \
    X(4, EC_I16, int16_t, EC_F32, float)    \
    X(5, EC_F32, float, EC_I16, int16_t)    \
    X(6, EC_F32, float, EC_F64, double)     \
    X(7, EC_F64, double, EC_F32, float)     \
    X(8, EC_U8, uint8_t, EC_F32, float)     \
    X(9, EC_F32, float, EC_U8, uint8_t)     \
    X(10, EC_U16, uint16_t, EC_I16, int16_t) \
    X(11, EC_I16, int16_t, EC_U16, uint16_t)
constexpr int kFusedMixedPairs = 12;

// slot patterns: bit k set = slot k (x, y, z, w) has type B
constexpr int kPatABAB = 0b1010;  // four operands
constexpr int kPatAAB = 0b0100;   // three operands (w unused)
constexpr int kPatABA = 0b0010;
constexpr int kPatABB = 0b0110;

template <typename Own, typename C>
__device__ __forceinline__ cells<Own, 2> same_or(const cells<Own, 2>& own, const cells<C, 2>& cand, bool take) {
    if constexpr (std::is_same<Own, C>::value) return take ? cand : own;
    else { (void)cand; (void)take; return own; }  // the host never aliases slots of different cell types
}

template <typename A, typename B, int PAT, int O1, int O2, int O3>
__global__ __launch_bounds__(kBlock) void k_fused_mixed(FusedArgs fa, double* __restrict__ out, uint8_t* __restrict__ out_mask, size_t n) {
    using TX = A;
    using TY = typename std::conditional<(PAT & 2) != 0, B, A>::type;
    using TZ = typename std::conditional<(PAT & 4) != 0, B, A>::type;
    using TW = typename std::conditional<(PAT & 8) != 0, B, A>::type;
    using X2 = cells<TX, 2>;
    using Y2 = cells<TY, 2>;
    using Z2 = cells<TZ, 2>;
    using W2 = cells<TW, 2>;
    constexpr int kFusedU = fused_u(sizeof(A) < sizeof(B) ? sizeof(A) : sizeof(B));  // by the NARROWER type: u16 + f32 runs the same at 2 and 4, f32 + f64 wants 2
    const unsigned head = fa.head;
    const size_t npairs = (n - head) >> 1;
    constexpr size_t TILE = size_t(kBlock) * kFusedU;
    const size_t tile = two_front_tile();
    const size_t base = tile * TILE + threadIdx.x;
    constexpr bool has_w = O3 != kOpNone;
    D2* __restrict__ op = reinterpret_cast<D2*>(out + head);
    const TX* __restrict__ px = static_cast<const TX*>(fa.p[0]) + head;
    const TY* __restrict__ py = static_cast<const TY*>(fa.p[1]) + head;
    const TZ* __restrict__ pz = static_cast<const TZ*>(fa.p[2]) + head;
    const TW* __restrict__ pw = static_cast<const TW*>(fa.p[3]) + head;
    // launch-uniform operand configuration, resolved once per wave
    const bool ld_x = !fa.is_sc[0], ld_y = !fa.is_sc[1] && fa.alias[1] == 1, ld_z = !fa.is_sc[2] && fa.alias[2] == 2,
               ld_w = has_w && !fa.is_sc[3] && fa.alias[3] == 3;
    const bool full = tile * TILE + TILE <= npairs;
    constexpr bool kSmallShape = is_small_int<A>::value && is_small_int<B>::value && is_ndvi_shape<O1, O2, O3>::value;
    const bool small_ints = kSmallShape && !(fa.is_sc[0] | fa.is_sc[1] | fa.is_sc[2] | (has_w ? fa.is_sc[3] : 0));
    constexpr bool kNdviTypes = std::is_same<TZ, TX>::value && std::is_same<TW, TY>::value;
    const int fast = full ? fused_fast_config<has_w>(fa) : 0;
    if (fast == 1) {
        fused_fast_tile<TX, TY, TZ, TW, O1, O2, O3, kFusedU, false>(px, py, pz, pw, op, base);
    } else if (has_w && kNdviTypes && fast == 2) {
        if constexpr (has_w && kNdviTypes) fused_fast_tile<TX, TY, TZ, TW, O1, O2, O3, kFusedU, true>(px, py, pz, pw, op, base);
    } else {
    X2 x[kFusedU] = {};
    Y2 y[kFusedU] = {};
    Z2 z[kFusedU] = {};
    W2 w[kFusedU] = {};
#pragma unroll
    for (int j = 0; j < kFusedU; ++j) {
        const size_t pr = base + size_t(j) * kBlock;
        if (full || pr < npairs) {
            if (ld_x) x[j] = load_cells<true, TX, 2>(px + 2 * pr);
            if (ld_y) y[j] = load_cells<true, TY, 2>(py + 2 * pr);
            if (ld_z) z[j] = load_cells<true, TZ, 2>(pz + 2 * pr);
            if (ld_w) w[j] = load_cells<true, TW, 2>(pw + 2 * pr);
        }
    }
#pragma unroll
    for (int j = 0; j < kFusedU; ++j) {
        const size_t pr = base + size_t(j) * kBlock;
        if (full || pr < npairs) {
            // aliased slots (z == x, w == y for NDVI) were loaded once: take the earlier slot's registers
            const Y2 yy = same_or<TY, TX>(y[j], x[j], fa.alias[1] == 0);
            Z2 zz = same_or<TZ, TX>(z[j], x[j], fa.alias[2] == 0);
            zz = same_or<TZ, TY>(zz, yy, fa.alias[2] == 1);
            W2 ww = w[j];
            if constexpr (has_w) {
                ww = same_or<TW, TX>(ww, x[j], fa.alias[3] == 0);
                ww = same_or<TW, TY>(ww, yy, fa.alias[3] == 1);
                ww = same_or<TW, TZ>(ww, zz, fa.alias[3] == 2);
            }
            const D2 vx = fa.is_sc[0] ? D2{fa.sc[0], fa.sc[0]} : D2{to_f64(x[j][0]), to_f64(x[j][1])};
            const D2 vy = fa.is_sc[1] ? D2{fa.sc[1], fa.sc[1]} : D2{to_f64(yy[0]), to_f64(yy[1])};
            const D2 vz = fa.is_sc[2] ? D2{fa.sc[2], fa.sc[2]} : D2{to_f64(zz[0]), to_f64(zz[1])};
            D2 vw = vz;
            if constexpr (has_w) vw = fa.is_sc[3] ? D2{fa.sc[3], fa.sc[3]} : D2{to_f64(ww[0]), to_f64(ww[1])};
            D2 o;
            if (small_ints) {  // launch-uniform
                if constexpr (kSmallShape) {
                    o.x = ndvi_shape_small_int<O1, O3>(vx.x, vy.x, vz.x, vw.x);
                    o.y = ndvi_shape_small_int<O1, O3>(vx.y, vy.y, vz.y, vw.y);
                }
            } else {
                o.x = fused_cell_t<O1, O2, O3>(vx.x, vy.x, vz.x, vw.x);
                o.y = fused_cell_t<O1, O2, O3>(vx.y, vy.y, vz.y, vw.y);
            }
            nt_store(o, op + pr);
        }
    }
    }  // general tile
    if (blockIdx.x == 0 && threadIdx.x < 2) {  // the peeled head cell (lane 0) and the odd tail cell (lane 1)
        const bool do_it = threadIdx.x == 0 ? head != 0 : ((n - head) & 1) != 0;
        const size_t i = threadIdx.x == 0 ? 0 : n - 1;
        if (do_it)
            st_cell(fused_cell_t<O1, O2, O3>(operand_cell(fa, 0, i), operand_cell(fa, 1, i), operand_cell(fa, 2, i),
                                             has_w ? operand_cell(fa, 3, i) : 0.0), out + i);
    }
    fused_mask_phase(fa, out_mask, n);
}

}  // namespace ecd
