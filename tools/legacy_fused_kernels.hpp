// legacy_fused_kernels.hpp — the fused kernels of rounds 1-2, kept OUT of the library for tools/tune_fused_any.hip:
// k_fused_same<T,O1,O2,O3> (one cell type, compile-time op triple) and k_fused_mixed<A,B,PAT,O1,O2,O3> (two cell
// types, per-slot compile-time types).  k_fused_any replaced both (ec_fused_any.hpp).
#pragma once

#include <type_traits>

#include "ec_fused_kernels.hpp"

namespace ecd {

// compile-time ops: a wave-uniform `switch` per cell costs ≈30 % on the NDVI kernel (it serialises the
// four cells of a lane; tools/tune_fused.hip: 422 vs 546 Gcells/s), so the vector kernel is
// instantiated per op triple (4 x 4 x 5 = 80) and per cell type.
template <int O1, int O2, int O3>
__device__ __forceinline__ double fused_cell_t(double x, double y, double z, double w) {
    const double t1 = cell_op<O1, true>(x, y);
    double t2 = z;
    if constexpr (O3 != kOpNone) t2 = cell_op<O3, true>(z, w);
    return cell_op<O2, true>(t1, t2);
}

// Chains of the NDVI shape on cells of at most 16 bits: `(x ± y) / (z ± w)` or `(x ± y) / z`.  The sums and
// differences are exact integers in [-131070, 131070], no NaN can reach the divide, and div_small_int is proven
// bit-exact on that whole square as well (68,717,903,881 pairs, tools/div_small_check.hip) — so the chain needs
// 2 adds and the 6-instruction divide instead of three IEEE steps with their NaN fix-ups.
template <int O1, int O2, int O3>
struct is_ndvi_shape {
    static constexpr bool value = O2 == EC_DIV && (O1 == EC_ADD || O1 == EC_SUB) && (O3 == EC_ADD || O3 == EC_SUB || O3 == kOpNone);
};
template <int O1, int O3>
__device__ __forceinline__ double ndvi_shape_small_int(double x, double y, double z, double w) {
    const double t1 = O1 == EC_ADD ? x + y : x - y;
    const double t2 = O3 == kOpNone ? z : (O3 == EC_ADD ? z + w : z - w);
    return div_small_int(t1, t2);
}


// per-type tile depth the legacy kernels were tuned to (round 2)
constexpr int legacy_fused_u(size_t widest_cell_bytes) {
#ifdef EC_FUSED_U
    return EC_FUSED_U;
#else
    return widest_cell_bytes >= 4 ? 2 : 4;
#endif
}

// The two operand configurations almost every call has — every operand its own buffer, or the NDVI aliasing
// `(x o1 y) o2 (x o3 y)` — with no scalar operand, on a full tile: everything launch-uniform in the general tile
// (which slots load, which alias which, which are scalars, the per-pair bounds check) is known, so the tile is
// straight-line code like k_binop_direct's (the general tile of the NDVI kernel carries 85 selects and several hundred
// scalar instructions).  MEASURED AND NOT USED: on one box, same run (profiles/r02/tune_fused_fast_tiles.log) the
// straight-line tiles gave NDVI u16 0.779 against 0.774 for the general tile, but NDVI u16 + f32 0.762 against 0.772,
// config 3 0.764 against 0.778 and (a+b)*c on f32 0.782 against 0.797 — these kernels wait on HBM, not on their
// instruction count.  The code stays behind EC_FUSED_FAST_TILES (off) so the comparison can be repeated.
template <typename TX, typename TY, typename TZ, typename TW, int O1, int O2, int O3, int U, bool NDVI>
__device__ __forceinline__ void fused_fast_tile(const TX* __restrict__ px, const TY* __restrict__ py,
                                                const TZ* __restrict__ pz, const TW* __restrict__ pw,
                                                D2* __restrict__ op, size_t base) {
    constexpr bool has_w = O3 != kOpNone;
    constexpr bool kSmall = is_small_int<TX>::value && is_small_int<TY>::value && is_small_int<TZ>::value &&
                            (!has_w || is_small_int<TW>::value) && is_ndvi_shape<O1, O2, O3>::value;
    cells<TX, 2> x[U];
    cells<TY, 2> y[U];
    cells<TZ, 2> z[U] = {};
    cells<TW, 2> w[U] = {};
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const size_t pr = base + size_t(j) * kBlock;
        x[j] = load_cells<true, TX, 2>(px + 2 * pr);
        y[j] = load_cells<true, TY, 2>(py + 2 * pr);
        if constexpr (!NDVI) {
            z[j] = load_cells<true, TZ, 2>(pz + 2 * pr);
            if constexpr (has_w) w[j] = load_cells<true, TW, 2>(pw + 2 * pr);
        }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const D2 vx{to_f64(x[j][0]), to_f64(x[j][1])}, vy{to_f64(y[j][0]), to_f64(y[j][1])};
        D2 vz = vx, vw = vy;  // NDVI: z is x, w is y
        if constexpr (!NDVI) {
            vz = D2{to_f64(z[j][0]), to_f64(z[j][1])};
            if constexpr (has_w) vw = D2{to_f64(w[j][0]), to_f64(w[j][1])};
        }
        D2 o;
        if constexpr (kSmall) {
            o.x = ndvi_shape_small_int<O1, O3>(vx.x, vy.x, vz.x, vw.x);
            o.y = ndvi_shape_small_int<O1, O3>(vx.y, vy.y, vz.y, vw.y);
        } else {
            o.x = fused_cell_t<O1, O2, O3>(vx.x, vy.x, vz.x, vw.x);
            o.y = fused_cell_t<O1, O2, O3>(vx.y, vy.y, vz.y, vw.y);
        }
        nt_store(o, op + base + size_t(j) * kBlock);
    }
}

// launch-uniform: which straight-line tile, if any, serves this call (0 none, 1 all operands distinct, 2 NDVI aliasing)
template <bool HAS_W>
__device__ __forceinline__ int fused_fast_config(const FusedArgs& fa) {
#ifndef EC_FUSED_FAST_TILES  // OFF in the library: measured slower, see above (build-time A/B switch, tools/tune_fused2.hip)
    return 0;
#endif
    if (fa.is_sc[0] | fa.is_sc[1] | fa.is_sc[2] | (HAS_W ? fa.is_sc[3] : 0)) return 0;
    if (fa.alias[1] != 1) return 0;
    if (fa.alias[2] == 2 && (!HAS_W || fa.alias[3] == 3)) return 1;
    if (HAS_W && fa.alias[2] == 0 && fa.alias[3] == 1) return 2;
    return 0;
}

// One workgroup per tile of kBlock*fused_u(sizeof T) pairs, two-front order, as k_binop_direct.  All buffer
// operands have cell type T.
template <typename T, int O1, int O2, int O3>
__global__ __launch_bounds__(kBlock) void k_fused_same(FusedArgs fa, double* __restrict__ out, uint8_t* __restrict__ out_mask, size_t n) {
    using T2 = cells<T, 2>;  // 1-byte cells as a 16-bit word, so that the pair loads keep `nt` (ec_device.hpp)
    constexpr int kFusedU = legacy_fused_u(sizeof(T));
    const unsigned head = fa.head;
    const size_t npairs = (n - head) >> 1;
    constexpr size_t TILE = size_t(kBlock) * kFusedU;
    const size_t tile = two_front_tile();
    const size_t base = tile * TILE + threadIdx.x;
    constexpr bool has_w = O3 != kOpNone;
    D2* __restrict__ op = reinterpret_cast<D2*>(out + head);
    const T* __restrict__ px = static_cast<const T*>(fa.p[0]) + head;
    const T* __restrict__ py = static_cast<const T*>(fa.p[1]) + head;
    const T* __restrict__ pz = static_cast<const T*>(fa.p[2]) + head;
    const T* __restrict__ pw = static_cast<const T*>(fa.p[3]) + head;
    // launch-uniform operand configuration, resolved once per wave
    const bool ld_x = !fa.is_sc[0], ld_y = !fa.is_sc[1] && fa.alias[1] == 1, ld_z = !fa.is_sc[2] && fa.alias[2] == 2,
               ld_w = has_w && !fa.is_sc[3] && fa.alias[3] == 3;
    const bool full = tile * TILE + TILE <= npairs;  // every pair of the tile exists: no per-pair guards
    constexpr bool kSmallShape = is_small_int<T>::value && is_ndvi_shape<O1, O2, O3>::value;
    const bool small_ints = kSmallShape && !(fa.is_sc[0] | fa.is_sc[1] | fa.is_sc[2] | (has_w ? fa.is_sc[3] : 0));  // no scalar operand
    const int fast = full ? fused_fast_config<has_w>(fa) : 0;
    if (fast == 1) {
        fused_fast_tile<T, T, T, T, O1, O2, O3, kFusedU, false>(px, py, pz, pw, op, base);
    } else if (has_w && fast == 2) {
        fused_fast_tile<T, T, T, T, O1, O2, O3, kFusedU, true>(px, py, pz, pw, op, base);
    } else {
    T2 x[kFusedU] = {}, y[kFusedU] = {}, z[kFusedU] = {}, w[kFusedU] = {};
#pragma unroll
    for (int j = 0; j < kFusedU; ++j) {
        const size_t pr = base + size_t(j) * kBlock;
        if (full || pr < npairs) {
            if (ld_x) x[j] = load_cells<true, T, 2>(px + 2 * pr);
            if (ld_y) y[j] = load_cells<true, T, 2>(py + 2 * pr);
            if (ld_z) z[j] = load_cells<true, T, 2>(pz + 2 * pr);
            if (ld_w) w[j] = load_cells<true, T, 2>(pw + 2 * pr);
        }
    }
#pragma unroll
    for (int j = 0; j < kFusedU; ++j) {
        const size_t pr = base + size_t(j) * kBlock;
        if (full || pr < npairs) {
            const T2 yy = fa.alias[1] == 1 ? y[j] : x[j];
            const T2 zz = fa.alias[2] == 2 ? z[j] : (fa.alias[2] == 0 ? x[j] : yy);
            const T2 ww = !has_w ? zz : fa.alias[3] == 3 ? w[j] : (fa.alias[3] == 0 ? x[j] : fa.alias[3] == 1 ? yy : zz);
            const D2 vx = fa.is_sc[0] ? D2{fa.sc[0], fa.sc[0]} : D2{to_f64(x[j][0]), to_f64(x[j][1])};
            const D2 vy = fa.is_sc[1] ? D2{fa.sc[1], fa.sc[1]} : D2{to_f64(yy[0]), to_f64(yy[1])};
            const D2 vz = fa.is_sc[2] ? D2{fa.sc[2], fa.sc[2]} : D2{to_f64(zz[0]), to_f64(zz[1])};
            const D2 vw = fa.is_sc[3] ? D2{fa.sc[3], fa.sc[3]} : D2{to_f64(ww[0]), to_f64(ww[1])};
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
    constexpr int kFusedU = legacy_fused_u(sizeof(A) < sizeof(B) ? sizeof(A) : sizeof(B));  // by the NARROWER type: u16 + f32 runs the same at 2 and 4, f32 + f64 wants 2
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
