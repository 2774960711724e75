// ec_fused_any.hpp — the fused two-level expression  out = (x o1 y) o2 (z o3 w)  over operands of ANY mix of the ten
// cell types, in one pass (gfx950).  Round 3: fused coverage by construction instead of by enumeration.
//
// Rounds 1-2 instantiated one kernel per cell type and op triple (k_fused_same: 800) and, for operands of two cell
// types, one per (ordered type pair, slot pattern, op triple) for 12 chosen pairs (k_fused_mixed: 1,344); every other
// mix — 78 of the 90 ordered pairs, A A B B, three or four cell types — paid a convert pass per narrower operand through
// pooled temporaries (22 instead of 14 B/cell on NDVI u16 + f32, an allocation inside the call, no hipGraph capture).
// This one family replaces all of them (and runs faster than the specialised kernels did: ec_fused_kernels.hpp).
//
// What the memory system needs to know at compile time is only each operand stream's WIDTH: a lane's pair of cells is
// one 2-, 4-, 8- or 16-byte load whatever the cells mean.  So a slot has a compile-time LOAD CLASS (the cell width in
// bytes; 0 = the slot has no stream of its own: it is a scalar or the same buffer as an earlier slot) and everything
// else is launch-uniform run-time state read from scalar registers:
//   * the cell KIND inside the class (unsigned / signed / float) picks the widening to f64 — branch-free for 1- and
//     2-byte cells (sign extension as (v ^ s) - s with s = 0 or the sign bit), a wave-uniform branch around the whole
//     tile's conversions for 4- and 8-byte cells (v_cvt_f64_u32 / _i32 / _f32; the u64 / i64 sequences);
//   * the three ops are wave-uniform switches around the whole tile's cells (NOT per cell: a per-cell switch serialises a
//     lane's divides and cost 30 %, tools/tune_fused.hip), every arm the same rounded f64 step as the eager chain
//     (cell_op<OP, true>: the reference's per-step semantics and the x86 NaN rule);
//   * a class-0 slot takes an earlier slot's widened cells or its scalar by a launch-uniform select.
// 5^4 = 625 kernels cover every operand mix, every aliasing and every scalar placement: no convert pass, no allocation,
// graph-capturable for all of them (tests/test_gpu_instantiations.py walks all 10^2 type pairs and the three-type mixes).
// The widening is value-preserving (`unify` + `to_f64`, src/value.rs:103-107,207), so results are
// bit-identical to the eager chain and to the oracle's chain.
#pragma once

#include "ec_fused_kernels.hpp"

namespace ecd {

constexpr int kFusedClasses = 5;  // class index -> cell bytes
__host__ __device__ constexpr int fused_class_bytes(int idx) { return idx == 0 ? 0 : idx == 1 ? 1 : idx == 2 ? 2 : idx == 3 ? 4 : 8; }
inline int fused_class_index(size_t bytes) { return bytes == 0 ? 0 : bytes == 1 ? 1 : bytes == 2 ? 2 : bytes == 4 ? 3 : 4; }

// what a lane loads for one PAIR of cells of byte width C — always unsigned words (1-byte cells as <2 x i8> would
// lose the non-temporal flag, ec_device.hpp)
struct no_stream {};
template <int C> struct raw_pair;
template <> struct raw_pair<0> { using type = no_stream; };
template <> struct raw_pair<1> { using type = uint16_t; };
template <> struct raw_pair<2> { using type = uint32_t; };
template <> struct raw_pair<4> { using type = vec<uint32_t, 2>; };
template <> struct raw_pair<8> { using type = vec<uint32_t, 4>; };

// cell kinds (dtype code >> 2: EC_U8..EC_U64 = 0..3, EC_I8..EC_I64 = 4..7, EC_F32 / EC_F64 = 8, 9)
constexpr int kKindUnsigned = 0, kKindSigned = 1, kKindFloat = 2;

// Widen the NP loaded pairs of one slot to 2*NP doubles.  `kind` is launch-uniform.
template <int C, int NP>
__device__ __forceinline__ void widen_pairs(const typename raw_pair<C>::type (&raw)[NP], int kind, double (&d)[2 * NP]) {
    if constexpr (C == 1 || C == 2) {
        constexpr int kBits = 8 * C;
        constexpr uint32_t kCellMask = (1u << kBits) - 1u;
        const int32_t s = kind == kKindSigned ? int32_t(1u << (kBits - 1)) : 0;  // sign extension as (v ^ s) - s: exact for both kinds
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint32_t w = raw[j];
            const int32_t c0 = int32_t(w & kCellMask), c1 = int32_t(w >> kBits);
            d[2 * j] = static_cast<double>((c0 ^ s) - s);
            d[2 * j + 1] = static_cast<double>((c1 ^ s) - s);
        }
    } else if constexpr (C == 4) {
        if (kind == kKindFloat) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const uint32_t w0 = raw[j].x, w1 = raw[j].y;  // scalars first (a bit_cast of a vector element miscompiled once)
                d[2 * j] = static_cast<double>(__builtin_bit_cast(float, w0));
                d[2 * j + 1] = static_cast<double>(__builtin_bit_cast(float, w1));
            }
        } else if (kind == kKindSigned) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const uint32_t w0 = raw[j].x, w1 = raw[j].y;
                d[2 * j] = static_cast<double>(static_cast<int32_t>(w0));
                d[2 * j + 1] = static_cast<double>(static_cast<int32_t>(w1));
            }
        } else {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const uint32_t w0 = raw[j].x, w1 = raw[j].y;
                d[2 * j] = static_cast<double>(w0);
                d[2 * j + 1] = static_cast<double>(w1);
            }
        }
    } else if constexpr (C == 8) {
        if (kind == kKindFloat) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const uint64_t q0 = uint64_t(raw[j].x) | (uint64_t(raw[j].y) << 32), q1 = uint64_t(raw[j].z) | (uint64_t(raw[j].w) << 32);
                d[2 * j] = __builtin_bit_cast(double, q0);
                d[2 * j + 1] = __builtin_bit_cast(double, q1);
            }
        } else if (kind == kKindSigned) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const uint64_t q0 = uint64_t(raw[j].x) | (uint64_t(raw[j].y) << 32), q1 = uint64_t(raw[j].z) | (uint64_t(raw[j].w) << 32);
                d[2 * j] = static_cast<double>(static_cast<int64_t>(q0));      // Rust `as f64`: round to nearest even
                d[2 * j + 1] = static_cast<double>(static_cast<int64_t>(q1));
            }
        } else {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const uint64_t q0 = uint64_t(raw[j].x) | (uint64_t(raw[j].y) << 32), q1 = uint64_t(raw[j].z) | (uint64_t(raw[j].w) << 32);
                d[2 * j] = static_cast<double>(q0);
                d[2 * j + 1] = static_cast<double>(q1);
            }
        }
    }
}

// One op over the whole tile's cells: the switch is wave-uniform and OUTSIDE the cell loop.
template <int N>
__device__ __forceinline__ void apply_tile(int op, const double (&a)[N], const double (&b)[N], double (&r)[N]) {
    // the NaN rule once for the tile's cells, out of line (cell_op_n, ec_device.hpp) — not a lane-wise branch per cell
    if (op == EC_ADD) cell_op_n<EC_ADD, true, false, N>(a, b, r);
    else if (op == EC_SUB) cell_op_n<EC_SUB, true, false, N>(a, b, r);
    else if (op == EC_MUL) cell_op_n<EC_MUL, true, false, N>(a, b, r);
    else cell_op_n<EC_DIV, true, false, N>(a, b, r);
}

__host__ __device__ constexpr int fused_any_min_class(int a, int b, int c, int d) {
    int m = 16;
    if (a && a < m) m = a;
    if (b && b < m) m = b;
    if (c && c < m) m = c;
    if (d && d < m) m = d;
    return m == 16 ? 8 : m;
}

// FusedArgs as the same-type kernel uses them, read this way here:
//   dt[k]     cell type of slot k (kind = dt >> 2)
//   alias[k]  j < k: slot k is the same buffer (and type) as slot j;  k: its own stream
//   is_sc[k]  slot k is the scalar sc[k]
//   small     every slot is a buffer of ≤16-bit integer cells and the chain has the NDVI shape (x ± y) / (z ± w | z):
//             the 2-add + 6-instruction exact divide of ec_fused_kernels.hpp (proven on the whole operand square)
#ifndef EC_FUSED_CHUNKED
#define EC_FUSED_CHUNKED 0  // build-time A/B switch: 1 = the expression runs chunk by chunk with each chunk's store right behind it
#endif

// (x o1 y) o2 (z o3 w) for the 2 NP cells of NP loaded pairs per slot: widening by launch-uniform kind, class-0 slots filled from an earlier
// slot or a scalar, the ops as wave-uniform switches around the cells.
template <int CX, int CY, int CZ, int CW, int NP>
__device__ __forceinline__ void fused_any_cells(const FusedArgs& fa, const typename raw_pair<CX>::type (&rx)[NP], const typename raw_pair<CY>::type (&ry)[NP],
                                                const typename raw_pair<CZ>::type (&rz)[NP], const typename raw_pair<CW>::type (&rw)[NP], bool has_w,
                                                double (&o)[2 * NP]) {
    constexpr int NC = 2 * NP;
    double vx[NC], vy[NC];
    if constexpr (CX != 0) widen_pairs<CX, NP>(rx, fa.dt[0] >> 2, vx);
    else {
#pragma unroll
        for (int i = 0; i < NC; ++i) vx[i] = fa.sc[0];
    }
    if constexpr (CY != 0) widen_pairs<CY, NP>(ry, fa.dt[1] >> 2, vy);
    else {
        const bool from_x = !fa.is_sc[1];  // a class-0 slot that is not a scalar is an alias; slot 1 can only alias slot 0
#pragma unroll
        for (int i = 0; i < NC; ++i) vy[i] = from_x ? vx[i] : fa.sc[1];
    }
    if (fa.small) {  // launch-uniform: NDVI shape on ≤16-bit integer cells (no scalar operand)
        double vz[NC], vw[NC];
        if constexpr (CZ != 0) widen_pairs<CZ, NP>(rz, fa.dt[2] >> 2, vz);
        else {
            const int a = fa.alias[2];
#pragma unroll
            for (int i = 0; i < NC; ++i) vz[i] = a == 0 ? vx[i] : vy[i];
        }
        if constexpr (CW != 0) widen_pairs<CW, NP>(rw, fa.dt[3] >> 2, vw);
        else {
            const int a = fa.alias[3];
#pragma unroll
            for (int i = 0; i < NC; ++i) vw[i] = a == 0 ? vx[i] : a == 1 ? vy[i] : vz[i];
        }
        // x - y == x + (-y) exactly; the sign flips are launch-uniform
        const uint64_t f1 = fa.o1 == EC_SUB ? 0x8000000000000000ull : 0ull;
        const uint64_t f3 = fa.o3 == EC_SUB ? 0x8000000000000000ull : 0ull;
        double t1[NC], t2[NC], y[NC], e[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            t1[i] = vx[i] + bits_f64(f64_bits(vy[i]) ^ f1);
            t2[i] = has_w ? vz[i] + bits_f64(f64_bits(vw[i]) ^ f3) : vz[i];
        }
        div_small_int_nonzero_staged<NC>(t1, t2, o, y, e);  // the NC chains of six dependent instructions overlap
#pragma unroll
        for (int i = 0; i < NC; ++i) o[i] = t2[i] == 0.0 ? div_by_zero(t1[i]) : o[i];
    } else {
        double t1[NC], t2[NC];
        apply_tile<NC>(fa.o1, vx, vy, t1);
        double vz[NC];
        if constexpr (CZ != 0) widen_pairs<CZ, NP>(rz, fa.dt[2] >> 2, vz);
        else {
            const int a = fa.is_sc[2] ? 2 : fa.alias[2];
#pragma unroll
            for (int i = 0; i < NC; ++i) vz[i] = a == 0 ? vx[i] : a == 1 ? vy[i] : fa.sc[2];
        }
        if (has_w) {
            double vw[NC];
            if constexpr (CW != 0) widen_pairs<CW, NP>(rw, fa.dt[3] >> 2, vw);
            else {
                const int a = fa.is_sc[3] ? 3 : fa.alias[3];
#pragma unroll
                for (int i = 0; i < NC; ++i) vw[i] = a == 0 ? vx[i] : a == 1 ? vy[i] : a == 2 ? vz[i] : fa.sc[3];
            }
            apply_tile<NC>(fa.o3, vz, vw, t2);
        } else {
#pragma unroll
            for (int i = 0; i < NC; ++i) t2[i] = vz[i];
        }
        apply_tile<NC>(fa.o2, t1, t2, o);
    }
}

template <int CX, int CY, int CZ, int CW>
__global__ __launch_bounds__(kBlock) void k_fused_any(FusedArgs fa, double* __restrict__ out, uint8_t* __restrict__ out_mask, size_t n) {
    constexpr int U = fused_u(size_t(fused_any_min_class(CX, CY, CZ, CW)));  // pairs per lane per tile (2: ec_fused_kernels.hpp)
    constexpr int NC = 2 * U;
    const unsigned head = fa.head;
    const size_t npairs = (n - head) >> 1;
    constexpr size_t TILE = size_t(kBlock) * U;
    const size_t tile = two_front_tile();
    const size_t base = tile * TILE + threadIdx.x;
    const bool has_w = fa.o3 != kOpNone;
    const bool full = tile * TILE + TILE <= npairs;
    D2* __restrict__ op = reinterpret_cast<D2*>(out + head);
    // first cell of the pair grid of each stream
    const char* bx = static_cast<const char*>(fa.p[0]) + size_t(head) * CX;
    const char* by = static_cast<const char*>(fa.p[1]) + size_t(head) * CY;
    const char* bz = static_cast<const char*>(fa.p[2]) + size_t(head) * CZ;
    const char* bw = static_cast<const char*>(fa.p[3]) + size_t(head) * CW;

    typename raw_pair<CX>::type rx[U] = {};
    typename raw_pair<CY>::type ry[U] = {};
    typename raw_pair<CZ>::type rz[U] = {};
    typename raw_pair<CW>::type rw[U] = {};
    if (full) {
        // launch-uniform load policy (policy_arms, ec_device.hpp): the bits of the slots that HAVE a stream, packed
        constexpr int kStreams = (CX != 0) + (CY != 0) + (CZ != 0) + (CW != 0);
        constexpr int kBitY = (CX != 0), kBitZ = kBitY + (CY != 0), kBitW = kBitZ + (CZ != 0);
        unsigned packed = 0;
        if constexpr (CX != 0) packed |= (fa.cacheable & 1u);
        if constexpr (CY != 0) packed |= ((fa.cacheable >> 1) & 1u) << kBitY;
        if constexpr (CZ != 0) packed |= ((fa.cacheable >> 2) & 1u) << kBitZ;
        if constexpr (CW != 0) packed |= ((fa.cacheable >> 3) & 1u) << kBitW;
        policy_arms<kStreams>(packed, [&](auto bits) {
            constexpr unsigned B = decltype(bits)::value;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const size_t pr = base + size_t(j) * kBlock;
                if constexpr (CX != 0) rx[j] = load_vec<!(B & 1u)>(reinterpret_cast<const typename raw_pair<CX>::type*>(bx) + pr);
                if constexpr (CY != 0) ry[j] = load_vec<!((B >> kBitY) & 1u)>(reinterpret_cast<const typename raw_pair<CY>::type*>(by) + pr);
                if constexpr (CZ != 0) rz[j] = load_vec<!((B >> kBitZ) & 1u)>(reinterpret_cast<const typename raw_pair<CZ>::type*>(bz) + pr);
                if constexpr (CW != 0) rw[j] = load_vec<!((B >> kBitW) & 1u)>(reinterpret_cast<const typename raw_pair<CW>::type*>(bw) + pr);
            }
        });
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t pr = base + size_t(j) * kBlock;
            if (pr < npairs) {
                if constexpr (CX != 0) rx[j] = nt_load(reinterpret_cast<const typename raw_pair<CX>::type*>(bx) + pr);
                if constexpr (CY != 0) ry[j] = nt_load(reinterpret_cast<const typename raw_pair<CY>::type*>(by) + pr);
                if constexpr (CZ != 0) rz[j] = nt_load(reinterpret_cast<const typename raw_pair<CZ>::type*>(bz) + pr);
                if constexpr (CW != 0) rw[j] = nt_load(reinterpret_cast<const typename raw_pair<CW>::type*>(bw) + pr);
            }
        }
    }
    // Over the tile's 2 U cells at once.  (EC_FUSED_CHUNKED 1: chunk by chunk — the pair of cells of one 16-byte store — with each chunk's store
    // right behind it, the form that gained 0.5-1 % in k_binop_direct's short divide and in the built-in EVI kernel.  Here it gains nothing
    // that exceeds the run-to-run spread: NDVI u16 0.824 / 0.824 against 0.829 / 0.809, NDVI u16 + f32 0.8225 / 0.8255 against 0.8145 /
    // 0.8253, config 3 fused 0.744 / 0.770 against 0.767 / 0.766; profiles/r04/fixed_chunked_ab.md.)
#if EC_FUSED_CHUNKED
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const typename raw_pair<CX>::type x1[1] = {rx[j]};
        const typename raw_pair<CY>::type y1[1] = {ry[j]};
        const typename raw_pair<CZ>::type z1[1] = {rz[j]};
        const typename raw_pair<CW>::type w1[1] = {rw[j]};
        double o[2];
        fused_any_cells<CX, CY, CZ, CW, 1>(fa, x1, y1, z1, w1, has_w, o);
        const size_t pr = base + size_t(j) * kBlock;
        if (full || pr < npairs) nt_store(D2{o[0], o[1]}, op + pr);
    }
#else
    double o[NC];
    fused_any_cells<CX, CY, CZ, CW, U>(fa, rx, ry, rz, rw, has_w, o);
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const size_t pr = base + size_t(j) * kBlock;
        if (full || pr < npairs) nt_store(D2{o[2 * j], o[2 * j + 1]}, op + pr);
    }
#endif
    if (blockIdx.x == 0 && threadIdx.x < 2) {  // the peeled head cell (lane 0) and the odd tail cell (lane 1)
        const bool do_it = threadIdx.x == 0 ? head != 0 : ((n - head) & 1) != 0;
        const size_t i = threadIdx.x == 0 ? 0 : n - 1;
        if (do_it)
            st_cell(fused_cell(fa, operand_cell(fa, 0, i), operand_cell(fa, 1, i), operand_cell(fa, 2, i),
                               has_w ? operand_cell(fa, 3, i) : 0.0), out + i);
    }
    fused_mask_phase(fa, out_mask, n);
}

using FusedAnyKernel = void (*)(FusedArgs, double*, uint8_t*, size_t);

// kernel for load classes (cx fixed per translation unit; iy, iz, iw are class INDICES 0..4)
template <int CX>
FusedAnyKernel fused_any_kernel(int iy, int iz, int iw);

}  // namespace ecd
