// ec_device.hpp — per-cell device functions shared by every kernel (gfx950).
//
// The scalar semantics come from src/value.rs of the reference:
//   binop   value.rs:199-217  (l as f64) op (r as f64), always Float64
//   neg     value.rs:224-240
//   order   value.rs:248-265  ints natural, floats total_cmp
// NaN policy: results match what the reference produces on x86-64
// (its CI platform): an operand NaN propagates lhs-first and quieted, a
// generated NaN is the x86 default 0xFFF8000000000000.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "erased_cells.h"

namespace ecd {

template <typename T, int N>
using vec = T __attribute__((ext_vector_type(N)));

// with_ct! (src/lib.rs:85-101): X(dtype code, C type)
#define EC_WITH_CT(X)      \
    X(EC_U8, uint8_t)      \
    X(EC_U16, uint16_t)    \
    X(EC_U32, uint32_t)    \
    X(EC_U64, uint64_t)    \
    X(EC_I8, int8_t)       \
    X(EC_I16, int16_t)     \
    X(EC_I32, int32_t)     \
    X(EC_I64, int64_t)     \
    X(EC_F32, float)       \
    X(EC_F64, double)

template <typename T> struct is_fp { static constexpr bool value = false; };
template <> struct is_fp<float> { static constexpr bool value = true; };
template <> struct is_fp<double> { static constexpr bool value = true; };

constexpr uint64_t kNegQNaN = 0xFFF8000000000000ull;  // x86-64 SSE default NaN
constexpr uint64_t kQuietBit = 0x0008000000000000ull;

__device__ __forceinline__ uint64_t f64_bits(double d) { return __builtin_bit_cast(uint64_t, d); }
__device__ __forceinline__ double bits_f64(uint64_t b) { return __builtin_bit_cast(double, b); }

// Rust `as f64`: ints round to nearest even, f32 widens exactly.
template <typename T>
__device__ __forceinline__ double to_f64(T v) { return static_cast<double>(v); }

template <int OP>
__device__ __forceinline__ double apply_op(double a, double b) {
    if constexpr (OP == EC_ADD) return a + b;
    else if constexpr (OP == EC_SUB) return a - b;
    else if constexpr (OP == EC_MUL) return a * b;
    else return a / b;  // IEEE-correct v_div_scale/v_rcp/fma/v_div_fmas/v_div_fixup sequence
}

// One cell of cv_bin_op!.  FP_IN: either operand type is floating point (an
// operand can then be NaN/inf); with integer operands only 0/0 can make a NaN.
// Quotient of two operands that are integers of at most 16 bits (u8 / i8 / u16 / i16 cells widened to f64):
// v_rcp_f64, one Newton step, the quotient, its exact residual and one correction — 6 FP64 instructions
// instead of the 11 of the compiler's IEEE expansion (div_scale x2, rcp, 4 fma, mul, fma, div_fmas, div_fixup).
// Not an approximation: all 98304² operand pairs were tried against the IEEE expansion and agree bit for bit
// (tools/div_small_check.hip; tests/test_gpu_instantiations.py repeats it through the C ABI).  b == 0 gives what
// the reference's f64 divide gives: ±inf, and the x86 default NaN for 0/0.
template <typename T>
struct is_small_int { static constexpr bool value = !is_fp<T>::value && sizeof(T) <= 2; };

// the six instructions alone: the quotient for b != 0 (b == 0 gives a NaN here, replaced by the caller)
__device__ __forceinline__ double div_small_int_nonzero(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    const double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}
// the same six instructions for N independent quotients, stage by stage (the N dependency chains overlap)
template <int N>
__device__ __forceinline__ void div_small_int_nonzero_staged(const double (&a)[N], const double (&b)[N], double (&q)[N], double (&y)[N], double (&e)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = __builtin_amdgcn_rcp(b[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_fma(-b[i], y[i], 1.0);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = __builtin_fma(y[i], e[i], y[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = a[i] * y[i];
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_fma(-b[i], q[i], a[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = __builtin_fma(e[i], y[i], q[i]);
}
// what the reference's f64 divide gives for b == 0
__device__ __forceinline__ double div_by_zero(double a) {
    return a == 0.0 ? bits_f64(kNegQNaN) : (a > 0.0 ? __builtin_inf() : -__builtin_inf());
}
__device__ __forceinline__ double div_small_int(double a, double b) {
    const double q = div_small_int_nonzero(a, b);
    return b == 0.0 ? div_by_zero(a) : q;
}

template <int OP, bool FP_IN, bool SMALL_INT = false>
__device__ __forceinline__ double cell_op(double a, double b) {
    if constexpr (OP == EC_DIV && SMALL_INT && !FP_IN) return div_small_int(a, b);
    double res = apply_op<OP>(a, b);
    if constexpr (FP_IN) {
        if (__builtin_expect(res != res, 0)) {
            uint64_t ba = f64_bits(a), bb = f64_bits(b);
            uint64_t fix = (a != a) ? (ba | kQuietBit) : (b != b) ? (bb | kQuietBit) : kNegQNaN;
            res = bits_f64(fix);
        }
    } else if constexpr (OP == EC_DIV) {
        // branch-free: a select keeps the unrolled cells' divide chains interleavable
        res = (res != res) ? bits_f64(kNegQNaN) : res;
    }
    return res;
}

// cv_bin_op! for the N cells a lane holds for one store (N = 2: a 16-byte output slot), the NaN rule tested ONCE for all of them and
// handled out of line behind a wave-uniform branch.  cell_op<OP, true> tests every cell with a lane-wise branch (v_cmp_u_f64, s_and_saveexec,
// s_cbranch_execnz — per cell); in kernels that are bound by how long a workgroup lives those four branches per tile cost 5-12 % (buffer ∘
// scalar u8: 0.70 with them, 0.79 without, same tile, same loads and stores: profiles/r04/nan_rule_per_pair.md).  Same cells: a NaN result
// takes the first NaN operand quieted, the x86 default NaN when neither operand is one.
template <int OP, bool FP_IN, bool SMALL_INT, int N>
__device__ __forceinline__ void cell_op_n(const double (&a)[N], const double (&b)[N], double (&r)[N]) {
    if constexpr (!FP_IN) {
#pragma unroll
        for (int i = 0; i < N; ++i) r[i] = cell_op<OP, false, SMALL_INT>(a[i], b[i]);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) r[i] = apply_op<OP>(a[i], b[i]);
        bool nan = false;
#pragma unroll
        for (int i = 0; i + 1 < N; i += 2) nan = nan || __builtin_isunordered(r[i], r[i + 1]);
        if constexpr (N & 1) nan = nan || r[N - 1] != r[N - 1];
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(nan) != 0, 0)) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint64_t fix = (a[i] != a[i]) ? (f64_bits(a[i]) | kQuietBit) : (b[i] != b[i]) ? (f64_bits(b[i]) | kQuietBit) : kNegQNaN;
                r[i] = (r[i] != r[i]) ? bits_f64(fix) : r[i];
            }
        }
    }
}

// f32/f64::total_cmp keys (src/value.rs:260-261) and their integer siblings:
// every cell type maps to an int64 whose natural order is the reference order.
template <typename T>
__device__ __host__ __forceinline__ int64_t order_key(T v) {
    if constexpr (sizeof(T) == 8 && !is_fp<T>::value && T(-1) > T(0)) {
        return static_cast<int64_t>(static_cast<uint64_t>(v) ^ 0x8000000000000000ull);  // u64
    } else if constexpr (!is_fp<T>::value) {
        return static_cast<int64_t>(v);
    } else if constexpr (sizeof(T) == 4) {
        int32_t b = __builtin_bit_cast(int32_t, v);
        b ^= static_cast<int32_t>(static_cast<uint32_t>(b >> 31) >> 1);
        return b;
    } else {
        int64_t b = __builtin_bit_cast(int64_t, v);
        b ^= static_cast<int64_t>(static_cast<uint64_t>(b >> 63) >> 1);
        return b;
    }
}

template <typename T>
__device__ __host__ __forceinline__ T key_value(int64_t k) {
    if constexpr (sizeof(T) == 8 && !is_fp<T>::value && T(-1) > T(0)) {
        return static_cast<T>(static_cast<uint64_t>(k) ^ 0x8000000000000000ull);
    } else if constexpr (!is_fp<T>::value) {
        return static_cast<T>(k);
    } else if constexpr (sizeof(T) == 4) {
        int32_t b = static_cast<int32_t>(k);
        b ^= static_cast<int32_t>(static_cast<uint32_t>(b >> 31) >> 1);
        return __builtin_bit_cast(float, b);
    } else {
        int64_t b = k;
        b ^= static_cast<int64_t>(static_cast<uint64_t>(b >> 63) >> 1);
        return __builtin_bit_cast(double, b);
    }
}

__device__ __host__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Global loads/stores of cell vectors.  The pointee type is declared with alignment 1: gfx950 runs with
// unaligned global access enabled (the compiler therefore still emits ONE dword/dwordx2/dwordx4
// instruction), so the same vector kernel serves any cell offset — a row-block of a raster whose width
// is not a multiple of 16 cells, a window into a larger buffer — instead of dropping to cell-wise code.
template <typename V>
struct under_aligned {
    typedef V type __attribute__((aligned(1)));
};
template <typename V>
__device__ __forceinline__ V nt_load(const V* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const typename under_aligned<V>::type*>(p));
}
// Streaming stores.  EC_STORE_POLICY (build-time A/B switch, like EC_NT_STORE in ec_runtime.hpp):
//   1  `nt`      — what rounds 1-3 shipped
//   2  `sc1 nt`  — write-through at device scope: the L2 passes each 64-byte write on as it completes instead of keeping the line
//                  dirty and writing it back when it is evicted.  With every operand byte coming from HBM (rotating operand sets)
//                  the 3 B-read / 8 B-write mix runs at 0.82 of the HBM peak with these against 0.78 with `nt` alone, the same
//                  tile shape and loads (tools/tune_store.hip, profiles/r04/tune_store_v1.log, _v2.log).
// hipcc has no source form for `sc1` on a plain store (volatile gives `sc0 sc1` on a flat store and waits for it; an atomic store
// stops at 64 bits), so policy 2 is inline asm: one `global_store_*` by width, the address as a 64-bit VGPR pair.  `s_nop 1` behind
// the 16-byte form: on gfx940+ a VALU write of the data registers of a store of more than 64 bits needs two wait states, and
// the compiler's hazard recognizer does not look into asm (without it the sweep's pure-write variants stored garbage).  No memory
// clobber: no kernel of the library reads what it stores, so the scheduler stays free to keep later loads above the store.
//   3  `sc1 nt` for VALUE streams (the f64 results, converted cells: nt_store / st_cell), `nt` for MASK streams (mask_store): the
//      default.  Write-through pays where the output is most of the launch's bytes and leaves the L2 in 16-byte-per-lane stores;
//      the 1 B/cell mask outputs (mask_and, mask_not, mask_from_nodata: a third to a half of their launch's bytes) ran 1-2 % slower
//      with it (profiles/r04/store_policy_ab.md).
#ifndef EC_STORE_POLICY
#define EC_STORE_POLICY 3
#endif
template <typename V>
__device__ __forceinline__ void stream_store_asm(V v, V* p) {
    static_assert(sizeof(V) == 1 || sizeof(V) == 2 || sizeof(V) == 4 || sizeof(V) == 8 || sizeof(V) == 16, "store width");
    if constexpr (sizeof(V) == 16) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(vec<uint32_t, 4>, v)));
    } else if constexpr (sizeof(V) == 8) {
        asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(__builtin_bit_cast(vec<uint32_t, 2>, v)));
    } else if constexpr (sizeof(V) == 4) {
        asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(__builtin_bit_cast(uint32_t, v)));
    } else if constexpr (sizeof(V) == 2) {
        asm volatile("global_store_short %0, %1, off sc1 nt" ::"v"(p), "v"(uint32_t(__builtin_bit_cast(uint16_t, v))));
    } else {
        asm volatile("global_store_byte %0, %1, off sc1 nt" ::"v"(p), "v"(uint32_t(__builtin_bit_cast(uint8_t, v))));
    }
}
template <typename V>
__device__ __forceinline__ void nt_store(V v, V* p) {
#if EC_STORE_POLICY >= 2
    stream_store_asm(v, p);
#else
    __builtin_nontemporal_store(v, reinterpret_cast<typename under_aligned<V>::type*>(p));
#endif
}
// the mask output of a launch (1 byte per cell, moved as words)
template <typename V>
__device__ __forceinline__ void mask_store(V v, V* p) {
#if EC_STORE_POLICY == 2
    stream_store_asm(v, p);
#else
    __builtin_nontemporal_store(v, reinterpret_cast<typename under_aligned<V>::type*>(p));
#endif
}
template <typename V>
__device__ __forceinline__ V plain_load(const V* p) {
    return *reinterpret_cast<const typename under_aligned<V>::type*>(p);
}
template <typename V>
__device__ __forceinline__ void plain_store(V v, V* p) {
    *reinterpret_cast<typename under_aligned<V>::type*>(p) = v;
}

// One cell (the peeled head cell, the odd tail cell, the ragged tail of a tile grid): non-temporal like every other
// access to caller data, so that "every load and store of a streaming kernel carries nt" holds without exceptions
// (tools/isa_audit.py).  Scalar accesses keep the flag at every width.
template <typename T>
__device__ __forceinline__ T ld_cell(const T* p) { return __builtin_nontemporal_load(p); }
template <typename T>
__device__ __forceinline__ void st_cell(T v, T* p) {
#if EC_STORE_POLICY >= 2
    stream_store_asm(v, p);
#else
    __builtin_nontemporal_store(v, p);
#endif
}

// N cells of type T as one lane loads them.  Cells of 2 bytes and more are the typed vector.  1-BYTE cells travel as
// unsigned words (uint16_t, uint32_t, 2 or 4 dwords) and are picked apart with shifts: hipcc (ROCm 7.2) drops the
// non-temporal flag from loads of <N x i8> vectors when it legalises them — every u8 / i8 / mask stream of rounds 1-2
// was loaded with plain `global_load_ushort/dword/dwordx2/dwordx4` while all other streams carried `nt` — and a
// bit-cast of a loaded word to <N x i8> is folded back into such a load.  As words the loads keep `nt`
// (tools/isa_audit.py checks every full-tile load of the library; A/B in profiles/r03/nt_u8_ab.md).
template <int N> struct byte_words;
template <> struct byte_words<2> { using type = uint16_t; };
template <> struct byte_words<4> { using type = uint32_t; };
template <> struct byte_words<8> { using type = vec<uint32_t, 2>; };
template <> struct byte_words<16> { using type = vec<uint32_t, 4>; };

template <typename T, int N, bool BYTES = sizeof(T) == 1>
struct cells {  // 2-, 4-, 8-byte cells
    using rep = vec<T, N>;
    rep v;
    __device__ __forceinline__ T operator[](int k) const { return v[k]; }
};
template <typename T, int N>
struct cells<T, N, true> {  // 1-byte cells
    using rep = typename byte_words<N>::type;
    rep v;
    __device__ __forceinline__ T operator[](int k) const {
        uint32_t w;
        if constexpr (N <= 4) w = static_cast<uint32_t>(v);
        else w = v[k >> 2];
        // signed cells by shift-left / arithmetic-shift-right (one v_bfe_i32): a mask-and-truncate chain of i8 cells is
        // recognised as a <N x i8> sign extension and folded back into a vector load (ConvertFn<i8, i16> did)
        if constexpr (T(-1) < T(0)) return static_cast<T>(static_cast<int32_t>(w << (24 - 8 * (k & 3))) >> 24);
        else return static_cast<T>((w >> (8 * (k & 3))) & 0xffu);
    }
};
template <bool NT, typename T, int N>
__device__ __forceinline__ cells<T, N> load_cells(const T* first_cell) {
    using C = cells<T, N>;
    const typename C::rep* p = reinterpret_cast<const typename C::rep*>(first_cell);
    if constexpr (NT) return C{nt_load(p)};
    else return C{plain_load(p)};
}

// Load policy of a launch.  Non-temporal is right for a stream that is read once (measured on reads that miss every
// cache: 16 B/lane loads reach 0.865 of the HBM peak with nt against 0.76 with the default policy, which allocates in
// the 256 MiB Infinity Cache and evicts on the way).  But an operand that FITS the Infinity Cache is worth keeping
// there: the 256 MiB u8 operand of the 16384² divide, re-read by the next operator on it, is then served on-die (the
// divide: 0.86 of peak against 0.80 with the same operand fetched from HBM every time; profiles/r03/
// tune_nt_width_rotating.log) — and when it is not re-read the default policy costs that one stream ≈1 %.  So the HOST
// decides per launch and per operand stream (cache_plan, ec_runtime.hpp: smallest operands first, while they fit the
// cache together) and passes one bit per stream: set = default cache policy, clear = nt.
//
// policy_arms<NBITS>(policy, leaf) turns the NBITS launch-uniform policy bits into 2^NBITS straight-line ARMS and calls
// leaf(std::integral_constant<unsigned, BITS>) in the one that matches: the leaf issues ALL operand loads of the tile
// with compile-time policies.  Two things make that shape necessary.  (1) One arm holds every stream's loads: with
// a branch per stream the first use of stream A's cells is scheduled inside A's arms and waits for A before B's
// loads are issued.  (2) The arms differ only in the loads' `!nontemporal` metadata, and LLVM merges such twins
// (SimplifyCFG hoists / sinks "identical" instructions out of the arms and keeps only the metadata they share — the
// launch would then run default-policy loads whatever the host asked for; tools/isa_audit.py caught exactly that).  An
// asm statement that clobbers memory and carries the arm's number brackets every arm with a default-policy load: loads
// cannot move across it and no two brackets are identical, so the arms stay apart.  It emits a comment, no instruction.
template <int NBITS, unsigned ACC = 0, typename Leaf>
__device__ __forceinline__ void policy_arms(unsigned policy, Leaf&& leaf) {
    if constexpr (NBITS == 0) {
        if constexpr (ACC != 0) asm volatile("; load-policy arm %0" ::"n"(ACC) : "memory");
        leaf(std::integral_constant<unsigned, ACC>{});
        if constexpr (ACC != 0) asm volatile("; end of load-policy arm %0" ::"n"(ACC) : "memory");
    } else {
        if (policy & (1u << (NBITS - 1))) policy_arms<NBITS - 1, (ACC | (1u << (NBITS - 1)))>(policy, leaf);
        else policy_arms<NBITS - 1, ACC>(policy, leaf);
    }
}

}  // namespace ecd
