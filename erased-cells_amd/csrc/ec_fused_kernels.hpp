// ec_fused_kernels.hpp — fused two-level expression  out = (x o1 y) o2 (z o3 w)  in one pass (gfx950).
//
// the reference evaluates operator chains eagerly — `(&nir - &red) / (nir + red)`
// (src/gdal/rasterband.rs:148,178) is three passes with two f64 temporaries, `(buf + ones) * 2.0`
// (examples/masked.rs:12) two passes.  Every intermediate of the eager chain is already an f64
// rounded once per step (src/value.rs:207), so evaluating the same steps in registers gives
// bit-identical results while the temporaries never touch HBM:
//   NDVI on u16:      eager 12 + 12 + 24 = 48 B/cell  ->  fused 2 + 2 + 8 = 12 B/cell
//   (a+b)*c f32+mask: eager 19 + 23 = 42 B/cell       ->  fused 4+4+4 + 8 + 3+1 = 24 B/cell
//
// The vector kernel `k_fused_same<T,O1,O2,O3>` handles buffer operands of ONE cell type (NDVI on u16
// bands, the f32 chain of config 3), instantiated per type and op triple: typed loads, all of a tile's
// loads in flight before the first use.  Operands of mixed types are first widened to their common
// `CellType::union` (value-preserving: the reference's own `unify`, src/value.rs:103-107) by the
// convert kernel into pooled temporaries — a kernel with run-time typed loaders was tried and ran at
// 27 % of peak (the per-type branches serialise its loads), slower than convert + same-type fusion.
// Aliased operands (z == x, w == y for NDVI) are loaded once; scalars cost no stream.
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
    uint8_t small;           // k_fused_any only: NDVI shape over ≤16-bit integer buffers (ec_fused_any.hpp)
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

// Pairs per lane per tile.  The right depth follows the operand width (tools/tune_fused2.hip, profiles/r02/
// tune_fused2.log): 1- and 2-byte cells want 4 (NDVI u16: 0.774 of peak at 4, 0.727 at 2, 0.750 at 8), 4- and
// 8-byte cells want 2 ((a+b)*c on f32 with masks: 0.794 at 2, 0.777 at 4, 0.774 at 8) — about the same bytes in
// flight per lane either way.  EC_FUSED_U forces one depth for every type (the tuner's builds).
#ifdef EC_FUSED_U
constexpr int fused_u(size_t) { return EC_FUSED_U; }
#else
constexpr int fused_u(size_t widest_cell_bytes) { return widest_cell_bytes >= 4 ? 2 : 4; }
#endif

// mask phase: AND of the distinct operand masks (src/masked/masked_buffer.rs:333 applied per step),
// 16 mask bytes per lane
__device__ __forceinline__ void fused_mask_phase(const FusedArgs& fa, uint8_t* __restrict__ out_mask, size_t n) {
    if (fa.nmask > 0) {
        const size_t ngroups = n / 16;
        const size_t stride = size_t(gridDim.x) * kBlock;
        u32x4* __restrict__ om = reinterpret_cast<u32x4*>(out_mask);
        for (size_t g = size_t(blockIdx.x) * kBlock + threadIdx.x; g < ngroups; g += stride) {
            u32x4 acc = nt_load(reinterpret_cast<const u32x4*>(fa.m[0]) + g);
            for (int k = 1; k < fa.nmask; ++k) acc &= nt_load(reinterpret_cast<const u32x4*>(fa.m[k]) + g);
            nt_store(acc, om + g);
        }
        if (blockIdx.x == 0)
            for (size_t i = ngroups * 16 + threadIdx.x; i < n; i += kBlock) {
                uint8_t acc = ld_cell(fa.m[0] + i);
                for (int k = 1; k < fa.nmask; ++k) acc &= ld_cell(fa.m[k] + i);
                st_cell(acc, out_mask + i);
            }
    }
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
    constexpr int kFusedU = fused_u(sizeof(T));
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
