// ec_map_kernels.hpp — the one-pass, per-cell kernels other than the binary
// arithmetic: neg, convert, fill, mask build/select/and/or/not (gfx950).
//
// All of them are pure streaming maps with different cell widths on the two
// sides.  One skeleton serves them: the widest stream moves 16 B per lane per
// access (CPL = 16 / widest cell cells per lane); the narrower streams then move
// CPL*width bytes per lane, still lane-contiguous, so every wave instruction
// touches one contiguous span.  A block tile is U such groups per lane, loads
// first, then compute + stores; one workgroup per tile, straight-line, with
// non-temporal loads and stores (every byte is touched once).  Ragged tails are
// guarded in the last tile.  Pointers may sit at any cell offset (under-aligned
// accesses, ec_device.hpp); the cell-wise kernel (one cell per lane) runs only
// when the "unaligned_vector" knob is off and a pointer is not 16-B aligned.
#pragma once

#include "ec_binop_kernels.hpp"

namespace ecd {

// Fn interface:
//   static constexpr int CPL;            cells per lane-group
//   typename In;                         what load() returns
//   template <bool NT0, bool NT1> In load(size_t g) const;   loads for group g (cells [g*CPL, g*CPL+CPL)); NT0 / NT1:
//                                        non-temporal (true) or default cache policy for the Fn's first / second
//                                        input stream (a Fn with one input ignores NT1, one with none both)
//   void store(size_t g, const In&) const;  compute + store for group g
//   void cell(size_t i) const;           one cell, any alignment
// `cacheable`: the launch's load policy (cache_plan, ec_runtime.hpp; policy_arms, ec_device.hpp) — bit 0 / bit 1 = the
// first / second input stream fits the Infinity Cache and is loaded with the default policy (policy_arms, ec_device.hpp).
template <typename Fn, int U>
__global__ __launch_bounds__(kBlock) void k_map(Fn fn, size_t n, unsigned cacheable) {
    using In = typename Fn::In;
    constexpr size_t CPL = Fn::CPL;
    const size_t ngroups = n / CPL;
    constexpr size_t TILE = size_t(kBlock) * U;
    const size_t tile = two_front_tile();  // one workgroup per tile, straight-line
    const size_t base = tile * TILE + threadIdx.x;
    if (tile * TILE + TILE <= ngroups) {
        In x[U];
        constexpr int kStreams = (Fn::kIn0 != 0) + (Fn::kIn1 != 0);  // input streams of the Fn (a second one only after a first)
        policy_arms<kStreams>(cacheable, [&](auto bits) {
            constexpr unsigned B = decltype(bits)::value;
#pragma unroll
            for (int j = 0; j < U; ++j) x[j] = fn.template load<!(B & 1u), !(B & 2u)>(base + size_t(j) * kBlock);
        });
#pragma unroll
        for (int j = 0; j < U; ++j) fn.store(base + size_t(j) * kBlock, x[j]);
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t g = base + size_t(j) * kBlock;
            if (g < ngroups) fn.store(g, fn.template load<true, true>(g));
        }
    }
    if (blockIdx.x == 0)
        for (size_t i = ngroups * CPL + threadIdx.x; i < n; i += kBlock) fn.cell(i);
}

template <typename Fn>
__global__ __launch_bounds__(kBlock) void k_map_cellwise(Fn fn, size_t n) {
    const size_t stride = size_t(gridDim.x) * kBlock;
    for (size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) fn.cell(i);
}

template <int A, int B>
struct max_of { static constexpr int value = A > B ? A : B; };

// ---- convert: BufferOps::convert (src/buffer.rs:161-164), per cell
// CellValue::convert (src/value.rs:74-98) == Rust `as` for every legal pair.
template <typename S, typename D>
struct ConvertFn {
    static constexpr int CPL = 16 / max_of<sizeof(S), sizeof(D)>::value;
    static constexpr size_t kIn0 = sizeof(S), kIn1 = 0;  // bytes per cell of the first / second input stream (load policy)
    using DV = vec<D, CPL>;
    using In = cells<S, CPL>;  // 1-byte sources travel as words: their loads keep `nt` (ec_device.hpp)
    const S* __restrict__ src;
    D* __restrict__ dst;
    template <bool NT0, bool NT1>
    __device__ __forceinline__ In load(size_t g) const { return load_cells<NT0, S, CPL>(src + g * CPL); }
    __device__ __forceinline__ void store(size_t g, const In& x) const {
        if constexpr (sizeof(S) == 1) {
            DV o;
#pragma unroll
            for (int k = 0; k < CPL; ++k) o[k] = static_cast<D>(x[k]);
            nt_store(o, reinterpret_cast<DV*>(dst) + g);
        } else {
            nt_store(__builtin_convertvector(x.v, DV), reinterpret_cast<DV*>(dst) + g);
        }
    }
    __device__ __forceinline__ void cell(size_t i) const { st_cell(static_cast<D>(ld_cell(src + i)), dst + i); }
};

// ---- neg: impl Neg for CellValue (src/value.rs:224-240)
template <typename T> struct NegOut { using type = T; };
template <> struct NegOut<uint8_t> { using type = int16_t; };
template <> struct NegOut<uint16_t> { using type = int32_t; };
template <> struct NegOut<uint32_t> { using type = double; };
template <> struct NegOut<uint64_t> { using type = double; };

template <typename T>
__device__ __forceinline__ typename NegOut<T>::type neg_cell(T v) {
    using O = typename NegOut<T>::type;
    if constexpr (is_fp<T>::value) {
        return -v;  // sign-bit flip, NaN included
    } else if constexpr (is_fp<O>::value) {
        return -static_cast<double>(v);  // u32/u64: -(v as f64)
    } else if constexpr (sizeof(O) > sizeof(T)) {
        return static_cast<O>(-static_cast<O>(v));  // u8 -> i16, u16 -> i32
    } else {
        using UT = typename NegOut<T>::type;  // signed: wrap at MIN like a release build
        if constexpr (sizeof(T) == 8) return static_cast<UT>(0ull - static_cast<uint64_t>(v));
        else return static_cast<UT>(0u - static_cast<uint32_t>(v));
    }
}

template <typename T>
struct NegFn {
    using O = typename NegOut<T>::type;
    static constexpr int CPL = 16 / max_of<sizeof(T), sizeof(O)>::value;
    static constexpr size_t kIn0 = sizeof(T), kIn1 = 0;
    using DV = vec<O, CPL>;
    using In = cells<T, CPL>;
    const T* __restrict__ src;
    O* __restrict__ dst;
    template <bool NT0, bool NT1>
    __device__ __forceinline__ In load(size_t g) const { return load_cells<NT0, T, CPL>(src + g * CPL); }
    __device__ __forceinline__ void store(size_t g, const In& x) const {
        DV o;
#pragma unroll
        for (int k = 0; k < CPL; ++k) o[k] = neg_cell<T>(x[k]);
        nt_store(o, reinterpret_cast<DV*>(dst) + g);
    }
    __device__ __forceinline__ void cell(size_t i) const { st_cell(neg_cell<T>(ld_cell(src + i)), dst + i); }
};

// ---- fill: BufferOps::fill (src/buffer.rs:79-88). W = cell width in bytes.
template <typename W>
struct FillFn {
    static constexpr int CPL = 16 / sizeof(W);
    static constexpr size_t kIn0 = 0, kIn1 = 0;
    using DV = vec<W, CPL>;
    struct In {};
    W* __restrict__ dst;
    W value;
    template <bool NT0, bool NT1>
    __device__ __forceinline__ In load(size_t) const { return In{}; }
    __device__ __forceinline__ void store(size_t g, const In&) const {
        DV o;
#pragma unroll
        for (int k = 0; k < CPL; ++k) o[k] = value;
        nt_store(o, reinterpret_cast<DV*>(dst) + g);
    }
    __device__ __forceinline__ void cell(size_t i) const { st_cell(value, dst + i); }
};

// ---- mask_from_nodata: from_vec_with_nodata (src/masked/masked_buffer.rs:62-71);
// equality under the total order is bit equality (src/masked/nodata.rs:42-49 ->
// src/value.rs:248-271). W = unsigned type of the cell width.
template <typename W>
struct MaskFromNodataFn {
    static constexpr int CPL = 16 / sizeof(W);
    static constexpr size_t kIn0 = sizeof(W), kIn1 = 0;
    using MV = vec<uint8_t, CPL>;
    using In = cells<W, CPL>;
    const W* __restrict__ src;
    uint8_t* __restrict__ mask;
    W nd;
    template <bool NT0, bool NT1>
    __device__ __forceinline__ In load(size_t g) const { return load_cells<NT0, W, CPL>(src + g * CPL); }
    __device__ __forceinline__ void store(size_t g, const In& x) const {
        MV m;
#pragma unroll
        for (int k = 0; k < CPL; ++k) m[k] = x[k] != nd;
        mask_store(m, reinterpret_cast<MV*>(mask) + g);
    }
    __device__ __forceinline__ void cell(size_t i) const { st_cell<uint8_t>(ld_cell(src + i) != nd, mask + i); }
};

// ---- mask_select: to_vec_with_nodata (src/masked/masked_buffer.rs:143-148)
template <typename W>
struct MaskSelectFn {
    static constexpr int CPL = 16 / sizeof(W);
    static constexpr size_t kIn0 = sizeof(W), kIn1 = 1;
    using SV = vec<W, CPL>;
    struct In { cells<W, CPL> x; cells<uint8_t, CPL> m; };  // the mask bytes as words (ec_device.hpp)
    const W* __restrict__ src;
    const uint8_t* __restrict__ mask;
    W* __restrict__ dst;
    W nd;
    template <bool NT0, bool NT1>
    __device__ __forceinline__ In load(size_t g) const {
        return In{load_cells<NT0, W, CPL>(src + g * CPL), load_cells<NT1, uint8_t, CPL>(mask + g * CPL)};
    }
    __device__ __forceinline__ void store(size_t g, const In& in) const {
        SV o;
#pragma unroll
        for (int k = 0; k < CPL; ++k) o[k] = in.m[k] ? in.x[k] : nd;
        nt_store(o, reinterpret_cast<SV*>(dst) + g);
    }
    __device__ __forceinline__ void cell(size_t i) const {
        const W v = ld_cell(src + i);  // unconditional: a load inside the select's arm comes out without `nt`
        st_cell<W>(ld_cell(mask + i) ? v : nd, dst + i);
    }
};

// ---- Mask BitAnd / BitOr / Not (src/masked/mask.rs:103-163): bytes are 0/1.
template <int KIND>  // 0 and, 1 or
struct MaskBin {
    static constexpr int CPL = 16;
    static constexpr size_t kIn0 = 1, kIn1 = 1;
    struct In { u32x4 a, b; };
    // no __restrict__: the owned forms run in place (out == l; BitAnd/BitOr for Mask, mask.rs:118-127,142-151).
    // Every lane loads a group and stores the same group, so aliasing is well defined without it.
    const uint8_t* l;
    const uint8_t* r;
    uint8_t* out;
    template <bool NT0, bool NT1>
    __device__ __forceinline__ In load(size_t g) const {
        return In{load_vec<NT0>(reinterpret_cast<const u32x4*>(l) + g), load_vec<NT1>(reinterpret_cast<const u32x4*>(r) + g)};
    }
    __device__ __forceinline__ void store(size_t g, const In& in) const {
        mask_store(KIND == 0 ? (in.a & in.b) : (in.a | in.b), reinterpret_cast<u32x4*>(out) + g);
    }
    __device__ __forceinline__ void cell(size_t i) const {
        const uint8_t a = ld_cell(l + i), b = ld_cell(r + i);
        st_cell<uint8_t>(KIND == 0 ? (a & b) : (a | b), out + i);
    }
};

struct MaskNot {
    static constexpr int CPL = 16;
    static constexpr size_t kIn0 = 1, kIn1 = 0;
    using In = u32x4;
    const uint8_t* m;  // may alias out (Not for Mask, mask.rs:103-109)
    uint8_t* out;
    template <bool NT0, bool NT1>
    __device__ __forceinline__ In load(size_t g) const { return load_vec<NT0>(reinterpret_cast<const u32x4*>(m) + g); }
    __device__ __forceinline__ void store(size_t g, const In& x) const {
        mask_store(x ^ 0x01010101u, reinterpret_cast<u32x4*>(out) + g);
    }
    __device__ __forceinline__ void cell(size_t i) const { st_cell<uint8_t>(ld_cell(m + i) ^ 1, out + i); }
};

// ---- synthetic inputs for bench/tests: counter-based, no stored vectors.
template <typename T>
struct SynthFn {
    static constexpr int CPL = 16 / sizeof(T);
    static constexpr size_t kIn0 = 0, kIn1 = 0;
    using DV = vec<T, CPL>;
    struct In {};
    T* __restrict__ dst;
    uint64_t seed, base, span;
    double lo, width;
    __device__ __forceinline__ T gen(size_t i) const {
        uint64_t h = splitmix64(seed ^ (base + i));
        if constexpr (is_fp<T>::value) return static_cast<T>(lo + width * (double(h >> 11) * 0x1.0p-53));
        else return static_cast<T>(uint64_t(lo) + h % span);
    }
    template <bool NT0, bool NT1>
    __device__ __forceinline__ In load(size_t) const { return In{}; }
    __device__ __forceinline__ void store(size_t g, const In&) const {
        DV o;
#pragma unroll
        for (int k = 0; k < CPL; ++k) o[k] = gen(g * CPL + k);
        nt_store(o, reinterpret_cast<DV*>(dst) + g);
    }
    __device__ __forceinline__ void cell(size_t i) const { st_cell(gen(i), dst + i); }
};

struct SynthMaskFn {
    static constexpr int CPL = 16;
    static constexpr size_t kIn0 = 0, kIn1 = 0;
    using DV = vec<uint8_t, 16>;
    struct In {};
    uint8_t* __restrict__ dst;
    uint64_t seed, base;
    uint32_t pct;
    __device__ __forceinline__ uint8_t gen(size_t i) const { return splitmix64(seed ^ (base + i)) % 100 >= pct; }
    template <bool NT0, bool NT1>
    __device__ __forceinline__ In load(size_t) const { return In{}; }
    __device__ __forceinline__ void store(size_t g, const In&) const {
        DV o;
#pragma unroll
        for (int k = 0; k < 16; ++k) o[k] = gen(g * 16 + k);
        nt_store(o, reinterpret_cast<DV*>(dst) + g);
    }
    __device__ __forceinline__ void cell(size_t i) const { st_cell(gen(i), dst + i); }
};

}  // namespace ecd
