// ec_reduce_kernels.hpp — min/max and mask counts (gfx950).
//
//   min_max  BufferOps::min_max, src/buffer.rs:169-173 (masked:
//            src/masked/masked_buffer.rs:208-217), order src/value.rs:248-265,
//            sentinels (T::MAX, T::MIN) src/ctype.rs:158-179 — finite for floats.
//   counts   Mask::counts, src/masked/mask.rs:72-80.
//
// Plan: read-only stream at 16 B per lane, U loads in flight per lane; each lane
// folds into register accumulators (packed 16-bit min/max for u16/i16), then a
// wavefront shuffle-reduce (DPP/ds_bpermute via __shfl_xor, 6 steps), then the
// four waves of the block combine through 64 B of LDS and lane 0 writes the
// block's partial.  A one-block finalize kernel folds the partials (≤ 8 per CU)
// and writes the two order-preserving int64 keys {~key(min), key(max)} — the
// form a MAX all-reduce over shards needs.  No atomics, so the result is
// deterministic and needs no zero-initialised memory.
#pragma once

#include <type_traits>

#include "ec_binop_kernels.hpp"

namespace ecd {

template <typename T> struct Limits;
#define EC_LIM(T, LO, HI) \
    template <> struct Limits<T> { static constexpr T lo = LO; static constexpr T hi = HI; };
EC_LIM(uint8_t, 0, UINT8_MAX)
EC_LIM(uint16_t, 0, UINT16_MAX)
EC_LIM(uint32_t, 0, UINT32_MAX)
EC_LIM(uint64_t, 0, UINT64_MAX)
EC_LIM(int8_t, INT8_MIN, INT8_MAX)
EC_LIM(int16_t, INT16_MIN, INT16_MAX)
EC_LIM(int32_t, INT32_MIN, INT32_MAX)
EC_LIM(int64_t, INT64_MIN, INT64_MAX)
EC_LIM(float, -3.402823466e+38f, 3.402823466e+38f)
EC_LIM(double, -1.7976931348623157e+308, 1.7976931348623157e+308)
#undef EC_LIM

// The type the per-lane accumulators live in: integers as themselves, floats as
// their total_cmp keys (same width, signed), so min/max are plain integer ops.
template <typename T> struct AccT { using type = T; };
template <> struct AccT<float> { using type = int32_t; };
template <> struct AccT<double> { using type = int64_t; };

template <typename T>
__device__ __forceinline__ typename AccT<T>::type acc_key(T v) {
    if constexpr (is_fp<T>::value) return static_cast<typename AccT<T>::type>(order_key<T>(v));
    else return v;
}

template <typename A>
__device__ __forceinline__ int64_t acc_to_i64(A a) {
    if constexpr (sizeof(A) == 8 && A(-1) > A(0)) return static_cast<int64_t>(static_cast<uint64_t>(a) ^ 0x8000000000000000ull);
    else return static_cast<int64_t>(a);
}

__device__ __forceinline__ int64_t wave_min_i64(int64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int64_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int64_t wave_max_i64(int64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int64_t o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// ---- 1-byte cells: gfx950 has no packed 8-bit min/max, so the four cells of a dword are folded as two packed
// 16-bit pairs with the cell in the HIGH byte of each 16-bit lane: a 16-bit compare is decided by its high byte, so
// the odd bytes of a dword need no unpacking at all (the dword itself is {odd:even, odd:even} = two lanes whose
// high bytes are the odd cells) and the even bytes need one packed shift left by 8.  Whatever sits in the low
// bytes only breaks ties between equal cells.  5 VALU ops per dword (1 shift + 2 x v_pk_min + 2 x v_pk_max)
// against 7 for unpack-to-16-bit-values; signed cells use the signed packed ops (the sign lives in the high byte).
template <typename T>
struct ByteFold {
    static constexpr bool kSigned = T(-1) < T(0);
    using H = typename std::conditional<kSigned, int16_t, uint16_t>::type;
    using H2 = vec<H, 2>;
    using U2 = vec<uint16_t, 2>;
    H2 mn_e, mn_o, mx_e, mx_o;  // even-byte and odd-byte accumulators, cell in the high byte

    static constexpr uint16_t kHiMin = uint16_t((uint16_t(uint8_t(Limits<T>::hi)) << 8) | 0xFFu);  // sentinel of a min fold
    static constexpr uint16_t kLoMax = uint16_t(uint16_t(uint8_t(Limits<T>::lo)) << 8);            // sentinel of a max fold

    __device__ __forceinline__ void init() {
        const H hi = static_cast<H>(kHiMin), lo = static_cast<H>(kLoMax);
        mn_e = mn_o = H2{hi, hi};
        mx_e = mx_o = H2{lo, lo};
    }
    static __device__ __forceinline__ H2 even(uint32_t x) { return __builtin_bit_cast(H2, __builtin_bit_cast(U2, x) << 8); }
    static __device__ __forceinline__ H2 odd(uint32_t x) { return __builtin_bit_cast(H2, x); }

    template <bool MASKED>
    __device__ __forceinline__ void fold(uint32_t x, uint32_t m) {
        H2 e = even(x), o = odd(x), e_mn = e, e_mx = e, o_mn = o, o_mx = o;
        if constexpr (MASKED) {
            // mask bytes are 0/1: widen each to a full 16-bit lane mask, then bit-select the sentinel
            const uint32_t me = __builtin_bit_cast(uint32_t, __builtin_bit_cast(U2, m & 0x00FF00FFu) * U2{0xFFFF, 0xFFFF});
            const uint32_t mo = __builtin_bit_cast(uint32_t, __builtin_bit_cast(U2, (m >> 8) & 0x00FF00FFu) * U2{0xFFFF, 0xFFFF});
            const uint32_t his = uint32_t(kHiMin) * 0x00010001u, los = uint32_t(kLoMax) * 0x00010001u;
            const uint32_t eb = __builtin_bit_cast(uint32_t, e), ob = __builtin_bit_cast(uint32_t, o);
            e_mn = __builtin_bit_cast(H2, (eb & me) | (his & ~me));
            e_mx = __builtin_bit_cast(H2, (eb & me) | (los & ~me));
            o_mn = __builtin_bit_cast(H2, (ob & mo) | (his & ~mo));
            o_mx = __builtin_bit_cast(H2, (ob & mo) | (los & ~mo));
        }
        mn_e = __builtin_elementwise_min(mn_e, e_mn);
        mx_e = __builtin_elementwise_max(mx_e, e_mx);
        mn_o = __builtin_elementwise_min(mn_o, o_mn);
        mx_o = __builtin_elementwise_max(mx_o, o_mx);
    }
    __device__ __forceinline__ T result_min() const {
        H2 a = __builtin_elementwise_min(mn_e, mn_o);
        const H h = a.x < a.y ? a.x : a.y;
        return static_cast<T>(h >> 8);  // arithmetic shift for signed cells, logical for unsigned
    }
    __device__ __forceinline__ T result_max() const {
        H2 a = __builtin_elementwise_max(mx_e, mx_o);
        const H h = a.x > a.y ? a.x : a.y;
        return static_cast<T>(h >> 8);
    }
};

// Launch shape of the reductions (tools/tune_reduce.hip, profiles/r02/tune_reduce.log): 512-thread workgroups with
// 8 x 16 B in flight per lane and a grid capped at 4 workgroups per CU read a 16384² byte mask in 41.4 µs including
// the finalize launch (0.81 of 8 TB/s); round 1's 256 threads x 4 loads x 8/CU took 43.3 µs, one tile per workgroup
// 42.4, and single-launch forms (write-through partials + relaxed ticket, the last workgroup folds) 41.6 at best.
constexpr int kRBlock = 512;
constexpr int kRWaves = kRBlock / kWave;

// partials[2*b] = min key, partials[2*b+1] = max key of block b (int64 order keys).
template <typename T, bool MASKED, int U, int BLOCK = kRBlock>
__global__ __launch_bounds__(BLOCK) void k_min_max_partials(const T* __restrict__ p, const uint8_t* __restrict__ mask,
                                                             size_t n, int64_t* __restrict__ partials, unsigned head,
                                                             int64_t* __restrict__ keys2_if_single) {
    // keys2_if_single != nullptr (only with a one-workgroup grid): this workgroup's fold IS the result, so it writes
    // {~key(min), key(max)} itself and the finalize launch is skipped — half the latency for small buffers.
    // `head` leading cells (reduce_head(), ec_runtime.hpp) are folded one by one by workgroup 0 so that the
    // 16-byte loads of the rest start 16-byte aligned: a window at an odd u16 offset otherwise reads 27 % slower.
    // Bits 8.. of `head`: the launch's load policy — bit 8 = the cells, bit 9 = the mask fit the Infinity Cache and are
    // loaded with the default cache policy instead of nt (cache_plan, ec_runtime.hpp; policy_arms, ec_device.hpp).
    const unsigned cacheable = head >> 8;
    head &= 0xffu;
    p += head;
    if constexpr (MASKED) mask += head;
    n -= head;
    using A = typename AccT<T>::type;
    constexpr int CPL = 16 / sizeof(T);
    using TV = cells<T, CPL>;        // 1-byte cells and the mask bytes travel as words: their loads keep `nt` (ec_device.hpp)
    using AV = vec<A, CPL>;
    using MV = cells<uint8_t, CPL>;
    constexpr bool BYTES = sizeof(T) == 1;
    const A hi0 = acc_key<T>(Limits<T>::hi), lo0 = acc_key<T>(Limits<T>::lo);
    AV vmin, vmax;
#pragma unroll
    for (int k = 0; k < CPL; ++k) { vmin[k] = hi0; vmax[k] = lo0; }
    ByteFold<typename std::conditional<BYTES, T, uint8_t>::type> bf;
    bf.init();

    const size_t ngroups = n / CPL;
    constexpr size_t TILE = size_t(BLOCK) * U;
    const size_t ntiles = (ngroups + TILE - 1) / TILE;
    auto fold = [&](const TV& x, const MV& m) {
        if constexpr (BYTES) {
#pragma unroll
            for (int k = 0; k < 4; ++k) bf.template fold<MASKED>(x.v[k], m.v[k]);
        } else {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                A key = acc_key<T>(x[k]);
                A kmin = key, kmax = key;
                if constexpr (MASKED) {
                    kmin = m[k] ? key : hi0;
                    kmax = m[k] ? key : lo0;
                }
                vmin[k] = kmin < vmin[k] ? kmin : vmin[k];
                vmax[k] = kmax > vmax[k] ? kmax : vmax[k];
            }
        }
    };

    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t base = tile * TILE + threadIdx.x;
        if (tile * TILE + TILE <= ngroups) {
            TV x[U];
            MV m[U] = {};
            policy_arms<(MASKED ? 2 : 1)>(cacheable, [&](auto bits) {  // bit 0: the cells, bit 1: the mask (ec_device.hpp)
                constexpr unsigned B = decltype(bits)::value;
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    x[j] = load_cells<!(B & 1u), T, CPL>(p + (base + size_t(j) * BLOCK) * CPL);
                    if constexpr (MASKED) m[j] = load_cells<!(B & 2u), uint8_t, CPL>(mask + (base + size_t(j) * BLOCK) * CPL);
                }
            });
#pragma unroll
            for (int j = 0; j < U; ++j) fold(x[j], m[j]);
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const size_t g = base + size_t(j) * BLOCK;
                if (g < ngroups) {
                    MV m = {};
                    if constexpr (MASKED) m = load_cells<true, uint8_t, CPL>(mask + g * CPL);
                    fold(load_cells<true, T, CPL>(p + g * CPL), m);
                }
            }
        }
    }
    // horizontal fold of the lane's accumulators, then the ragged tail cells
    A amin = vmin[0], amax = vmax[0];
    if constexpr (BYTES) {
        amin = bf.result_min();
        amax = bf.result_max();
    } else {
#pragma unroll
        for (int k = 1; k < CPL; ++k) {
            amin = vmin[k] < amin ? vmin[k] : amin;
            amax = vmax[k] > amax ? vmax[k] : amax;
        }
    }
    if (blockIdx.x == 0) {
        auto fold_cell = [&](ptrdiff_t i) {
            if (MASKED && !ld_cell(mask + i)) return;
            A key = acc_key<T>(ld_cell(p + i));
            amin = key < amin ? key : amin;
            amax = key > amax ? key : amax;
        };
        for (size_t i = ngroups * CPL + threadIdx.x; i < n; i += BLOCK) fold_cell(static_cast<ptrdiff_t>(i));
        for (unsigned h = threadIdx.x; h < head; h += BLOCK) fold_cell(-static_cast<ptrdiff_t>(h) - 1);  // the peeled cells
    }
    int64_t kmin = wave_min_i64(acc_to_i64<A>(amin));
    int64_t kmax = wave_max_i64(acc_to_i64<A>(amax));
    __shared__ int64_t s_min[(BLOCK / kWave)], s_max[(BLOCK / kWave)];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) { s_min[wave] = kmin; s_max[wave] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < (BLOCK / kWave); ++w) {
            kmin = s_min[w] < kmin ? s_min[w] : kmin;
            kmax = s_max[w] > kmax ? s_max[w] : kmax;
        }
        if (keys2_if_single) {
            keys2_if_single[0] = ~kmin;
            keys2_if_single[1] = kmax;
        } else {
            partials[2 * size_t(blockIdx.x)] = kmin;
            partials[2 * size_t(blockIdx.x) + 1] = kmax;
        }
    }
}

// Same reduction, cell-wise loads: any alignment.
template <typename T, bool MASKED>
__global__ __launch_bounds__(kBlock) void k_min_max_partials_cellwise(const T* __restrict__ p, const uint8_t* __restrict__ mask,
                                                                      size_t n, int64_t* __restrict__ partials) {
    using A = typename AccT<T>::type;
    A amin = acc_key<T>(Limits<T>::hi), amax = acc_key<T>(Limits<T>::lo);
    const size_t stride = size_t(gridDim.x) * kBlock;
    for (size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        if (MASKED && !mask[i]) continue;
        A key = acc_key<T>(p[i]);
        amin = key < amin ? key : amin;
        amax = key > amax ? key : amax;
    }
    int64_t kmin = wave_min_i64(acc_to_i64<A>(amin));
    int64_t kmax = wave_max_i64(acc_to_i64<A>(amax));
    __shared__ int64_t s_min[kWavesPerBlock], s_max[kWavesPerBlock];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) { s_min[wave] = kmin; s_max[wave] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) {
            kmin = s_min[w] < kmin ? s_min[w] : kmin;
            kmax = s_max[w] > kmax ? s_max[w] : kmax;
        }
        partials[2 * size_t(blockIdx.x)] = kmin;
        partials[2 * size_t(blockIdx.x) + 1] = kmax;
    }
}

// The finalize kernels are one workgroup of 1024 threads: with at most kFinalizeMaxParts partials every thread
// issues all its loads up front (≤ 4, independent), so the launch costs one memory latency instead of a chain
// of them — a 256-thread loop took 4.7 µs, a tenth of a 1 B/cell pass over 16384² cells.
constexpr int kFinalizeBlock = 1024;
constexpr int kFinalizeWaves = kFinalizeBlock / kWave;
constexpr int kFinalizeMaxParts = 4096;
constexpr int kFinalizePer = kFinalizeMaxParts / kFinalizeBlock;

// keys2 = {~min, max}: a MAX reduction over shards of both words is the global answer.
__global__ __launch_bounds__(kFinalizeBlock) void k_min_max_finalize(const int64_t* __restrict__ partials, int nparts,
                                                                      int64_t sentinel_min, int64_t sentinel_max,
                                                                      int64_t* __restrict__ keys2) {
    using K2 = vec<int64_t, 2>;
    const K2* __restrict__ pp = reinterpret_cast<const K2*>(partials);
    K2 v[kFinalizePer];
#pragma unroll
    for (int j = 0; j < kFinalizePer; ++j) {  // every load of the block is in flight before the first compare
        const int i = threadIdx.x + j * kFinalizeBlock;
        v[j] = i < nparts ? pp[i] : K2{sentinel_min, sentinel_max};
    }
    int64_t kmin = sentinel_min, kmax = sentinel_max;
#pragma unroll
    for (int j = 0; j < kFinalizePer; ++j) {
        kmin = v[j].x < kmin ? v[j].x : kmin;
        kmax = v[j].y > kmax ? v[j].y : kmax;
    }
    kmin = wave_min_i64(kmin);
    kmax = wave_max_i64(kmax);
    __shared__ int64_t s_min[kFinalizeWaves], s_max[kFinalizeWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) { s_min[wave] = kmin; s_max[wave] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kFinalizeWaves; ++w) {
            kmin = s_min[w] < kmin ? s_min[w] : kmin;
            kmax = s_max[w] > kmax ? s_max[w] : kmax;
        }
        keys2[0] = ~kmin;
        keys2[1] = kmax;
    }
}

// ---- first differing cell of two buffers (impl Ord / PartialEq for CellBuffer, src/buffer.rs:373-436).
// Equality under the reference's total order is bit equality for every cell type, so the scan works
// on raw cell words of width W; the ordering of the first differing pair is decided by the caller.
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint64_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

template <typename W, int U>
__global__ __launch_bounds__(kRBlock) void k_first_diff_partials(const W* __restrict__ l, const W* __restrict__ r, size_t n,
                                                                uint64_t* __restrict__ partials, bool aligned,
                                                                unsigned head) {
    constexpr int CPL = 16 / sizeof(W);
    using WV = cells<W, CPL>;
    uint64_t first = ~0ull;
    const unsigned cacheable = head >> 8;  // load policy of the launch: bit 8 = l, bit 9 = r (see k_min_max_partials)
    head &= 0xffu;
    if (aligned) {
        // `head` leading cells are compared singly by workgroup 0 (see k_min_max_partials); indices stay absolute
        if (blockIdx.x == 0)
            for (unsigned h = threadIdx.x; h < head; h += kRBlock)
                if (ld_cell(l + h) != ld_cell(r + h)) first = h < first ? h : first;
        l += head;
        r += head;
        n -= head;
        const size_t ngroups = n / CPL;
        constexpr size_t TILE = size_t(kRBlock) * U;
        const size_t ntiles = (ngroups + TILE - 1) / TILE;
        for (size_t tile = blockIdx.x; tile < ntiles && first == ~0ull; tile += gridDim.x) {
            const size_t base = tile * TILE + threadIdx.x;
            auto compare = [&](size_t g, const WV& a, const WV& b) {
                // the 16 bytes as four words first: equal groups (the common case, and all of a scan that ends in
                // "equal") cost 4 xor + 3 or instead of one compare per cell — 16 of them for 1-byte cells
                const u32x4 d = __builtin_bit_cast(u32x4, a.v) ^ __builtin_bit_cast(u32x4, b.v);
                if ((d.x | d.y | d.z | d.w) == 0) return;
#pragma unroll
                for (int k = CPL - 1; k >= 0; --k)
                    if (a[k] != b[k]) { const uint64_t i = head + g * CPL + k; first = i < first ? i : first; }
            };
            if (tile * TILE + TILE <= ngroups) {  // full tile: all 2 x U loads in flight before the first compare
                WV a[U], b[U];
                policy_arms<2>(cacheable, [&](auto bits) {
                    constexpr unsigned B = decltype(bits)::value;
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        a[j] = load_cells<!(B & 1u), W, CPL>(l + (base + size_t(j) * kRBlock) * CPL);
                        b[j] = load_cells<!(B & 2u), W, CPL>(r + (base + size_t(j) * kRBlock) * CPL);
                    }
                });
                // equal tiles (the common case, and all of a scan that ends in "equal") leave through ONE wave-uniform branch: the
                // tile's 2 U groups are xor-ed and or-ed together first and the wave votes; only a wave that holds a difference
                // looks at its groups one by one (before round 4: a lane-wise branch per group, eight per tile, in a kernel bound by
                // how long a workgroup lives — the NaN rule's lesson, profiles/r04/nan_rule_per_pair.md)
                u32x4 any = {0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < U; ++j) any |= __builtin_bit_cast(u32x4, a[j].v) ^ __builtin_bit_cast(u32x4, b[j].v);
                if (__builtin_amdgcn_ballot_w64((any.x | any.y | any.z | any.w) != 0) != 0) {
#pragma unroll
                    for (int j = 0; j < U; ++j) compare(base + size_t(j) * kRBlock, a[j], b[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const size_t g = base + size_t(j) * kRBlock;
                    if (g < ngroups) compare(g, load_cells<true, W, CPL>(l + g * CPL), load_cells<true, W, CPL>(r + g * CPL));
                }
            }
        }
        if (blockIdx.x == 0)
            for (size_t i = ngroups * CPL + threadIdx.x; i < n; i += kRBlock)
                if (ld_cell(l + i) != ld_cell(r + i)) first = head + i < first ? head + i : first;
    } else {
        const size_t stride = size_t(gridDim.x) * kRBlock;
        for (size_t i = size_t(blockIdx.x) * kRBlock + threadIdx.x; i < n && first == ~0ull; i += stride)
            if (ld_cell(l + i) != ld_cell(r + i)) first = i;
    }
    first = wave_min_u64(first);
    __shared__ uint64_t s_first[kRWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) s_first[wave] = first;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kRWaves; ++w) first = s_first[w] < first ? s_first[w] : first;
        partials[blockIdx.x] = first;
    }
}

__global__ __launch_bounds__(kFinalizeBlock) void k_first_diff_finalize(const uint64_t* __restrict__ partials, int nparts,
                                                                         uint64_t* __restrict__ result) {
    uint64_t v[kFinalizePer];
#pragma unroll
    for (int j = 0; j < kFinalizePer; ++j) {
        const int i = threadIdx.x + j * kFinalizeBlock;
        v[j] = i < nparts ? partials[i] : ~0ull;
    }
    uint64_t first = ~0ull;
#pragma unroll
    for (int j = 0; j < kFinalizePer; ++j) first = v[j] < first ? v[j] : first;
    first = wave_min_u64(first);
    __shared__ uint64_t s_first[kFinalizeWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) s_first[wave] = first;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kFinalizeWaves; ++w) first = s_first[w] < first ? s_first[w] : first;
        result[0] = first;
    }
}

// ---- Mask::counts: bytes are 0/1, so the count of true cells is the byte sum.
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int U>
__global__ __launch_bounds__(kRBlock) void k_mask_count_partials(const uint8_t* __restrict__ m, size_t n,
                                                                uint64_t* __restrict__ partials, bool aligned, unsigned head,
                                                                uint64_t* __restrict__ counts2_if_single, uint64_t* __restrict__ acc_or_null,
                                                                uint64_t* __restrict__ counts2_if_acc) {
    const uint64_t n_total = n;
    uint64_t cnt = 0;
    const bool cacheable = ((head >> 8) & 1u) != 0;  // load policy of the launch (see k_min_max_partials)
    head &= 0xffu;
    if (aligned) {
        if (blockIdx.x == 0)  // peeled leading cells (see k_min_max_partials)
            for (unsigned h = threadIdx.x; h < head; h += kRBlock) cnt += ld_cell(m + h) & 1;
        m += head;
        n -= head;
        const size_t ngroups = n / 16;
        constexpr size_t TILE = size_t(kRBlock) * U;
        const size_t ntiles = (ngroups + TILE - 1) / TILE;
        const u32x4* __restrict__ mv = reinterpret_cast<const u32x4*>(m);
        uint32_t c32 = 0;
        for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const size_t base = tile * TILE + threadIdx.x;
            auto pop16 = [](u32x4 x) {
                return __builtin_popcount(x.x & 0x01010101u) + __builtin_popcount(x.y & 0x01010101u) +
                       __builtin_popcount(x.z & 0x01010101u) + __builtin_popcount(x.w & 0x01010101u);
            };
            if (tile * TILE + TILE <= ngroups) {  // full tile: every load in flight before the first popcount
                u32x4 x[U];
                policy_arms<1>(cacheable ? 1u : 0u, [&](auto bits) {
#pragma unroll
                    for (int j = 0; j < U; ++j) x[j] = load_vec<!(decltype(bits)::value & 1u)>(mv + base + size_t(j) * kRBlock);
                });
#pragma unroll
                for (int j = 0; j < U; ++j) c32 += pop16(x[j]);
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const size_t g = base + size_t(j) * kRBlock;
                    if (g < ngroups) c32 += pop16(nt_load(mv + g));
                }
            }
            if (c32 > 0x7fff0000u) { cnt += c32; c32 = 0; }
        }
        cnt += c32;
        if (blockIdx.x == 0)
            for (size_t i = ngroups * 16 + threadIdx.x; i < n; i += kRBlock) cnt += ld_cell(m + i) & 1;
    } else {
        const size_t stride = size_t(gridDim.x) * kRBlock;
        for (size_t i = size_t(blockIdx.x) * kRBlock + threadIdx.x; i < n; i += stride) cnt += ld_cell(m + i) & 1;
    }
    cnt = wave_sum_u64(cnt);
    __shared__ uint64_t s_cnt[kRWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kRWaves; ++w) cnt += s_cnt[w];
        if (counts2_if_single) {  // one-workgroup grid: this IS the result (see k_min_max_partials)
            counts2_if_single[0] = cnt;
            counts2_if_single[1] = n_total - cnt;
        } else if (acc_or_null) {
            // One launch (round 4): ticket and sum share one 64-bit word — every workgroup adds (1 << 40 | its count) with ONE returning
            // device-scope atomic; the workgroup that finds grid - 1 tickets before it has just completed the sum, writes the result and
            // leaves the word at zero for the next kernel on this stream.  A sum is order-free: the result is deterministic.  (The host
            // passes acc only for n < 2^40; min / max do not pack with a ticket, so k_min_max_* keep their finalize launch.)
            const uint64_t old = __hip_atomic_fetch_add(acc_or_null, (1ull << 40) | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((old >> 40) == gridDim.x - 1u) {
                const uint64_t total = (old & ((1ull << 40) - 1)) + cnt;
                counts2_if_acc[0] = total;
                counts2_if_acc[1] = n_total - total;
                __hip_atomic_store(acc_or_null, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            partials[blockIdx.x] = cnt;
        }
    }
}

__global__ __launch_bounds__(kFinalizeBlock) void k_mask_count_finalize(const uint64_t* __restrict__ partials, int nparts,
                                                                         uint64_t n, uint64_t* __restrict__ counts2) {
    uint64_t v[kFinalizePer];
#pragma unroll
    for (int j = 0; j < kFinalizePer; ++j) {
        const int i = threadIdx.x + j * kFinalizeBlock;
        v[j] = i < nparts ? partials[i] : 0ull;
    }
    uint64_t cnt = 0;
#pragma unroll
    for (int j = 0; j < kFinalizePer; ++j) cnt += v[j];
    cnt = wave_sum_u64(cnt);
    __shared__ uint64_t s_cnt[kFinalizeWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kFinalizeWaves; ++w) cnt += s_cnt[w];
        counts2[0] = cnt;
        counts2[1] = n - cnt;
    }
}

}  // namespace ecd
