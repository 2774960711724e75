// ec_binop_kernels.hpp — element-wise  out[i] = f64(l[i]) op f64(r[i])  (gfx950).
//
// Replaces the iterator chain of src/buffer.rs:327 (+ src/value.rs:199-217 per
// cell) and its scalar-rhs sibling src/buffer.rs:346-352.
//
// Data layout / access plan: the output is always f64, so
// it is the widest stream (8 of the 11 B/cell for u8÷u16).  Work is cut into
// chunks of 128 cells = one `global_store_dwordx4` per lane per chunk (1 KiB,
// fully coalesced).  The narrow operand streams are moved in one of two ways:
//   DIRECT  lane-contiguous narrow loads: 2 cells per lane per chunk
//           (ushort / dword / dwordx2 / dwordx4 by operand width).
//   LDS     the wave loads its whole 256-cell tile of a ≤4-byte operand with ONE
//           load of 4*sizeof(T) bytes per lane into a wave-private LDS slab, then
//           each lane reads back the two cells that belong to each 16-B output
//           slot and widens them ("LDS staging for the widening step").  The slab
//           is private to the wave, so no workgroup barrier is needed.
// One workgroup per tile of U = 2 chunks per wave (1,024 cells), straight-line code,
// grid = number of tiles (≫ 256 CUs: 262,144 workgroups at 16384²), dealt from both
// ends of the buffer at once (two_front_tile).  Loads and stores are non-temporal:
// every byte is touched once and the streams (2.95 GB) dwarf the 256 MiB Infinity
// Cache.  There is no reuse between workgroups (each 128-B line is touched by exactly
// one wave), so an XCD-aware remap has no L2 locality to win (measured: −4 %).
#pragma once

#include "ec_device.hpp"

#ifndef EC_DIV_STAGE
#define EC_DIV_STAGE 2  // build-time A/B switch of the short divide's tile code: 1 = the tile's 2 U quotients staged together, 2 = chunk by chunk (default)
#endif

namespace ecd {

constexpr int kBlock = 256;          // 4 waves
constexpr int kWave = 64;
constexpr int kWavesPerBlock = kBlock / kWave;

template <bool NT, typename V>
__device__ __forceinline__ void store_vec(V* p, V v) {
    if constexpr (NT) nt_store(v, p);
    else plain_store(v, p);
}

template <bool NT, typename V>
__device__ __forceinline__ V load_vec(const V* p) {
    if constexpr (NT) return nt_load(p);
    else return plain_load(p);
}

// Workgroup -> tile map: even workgroups walk the buffer from the front, odd ones from the back, so
// two streaming fronts are live at once (measured +1…5 % over a single front, tune_binop_v4/v5.log).
// A bijection on [0, gridDim.x) for any grid size.
__device__ __forceinline__ size_t two_front_tile() {
    const size_t b = blockIdx.x;
    return (b & 1) ? size_t(gridDim.x) - 1 - (b >> 1) : (b >> 1);
}

// Where a lane's pairs sit inside its workgroup's tile of kBlock * U pairs.  EC_WAVE_CONTIG = 0: chunk j of all four waves is one
// contiguous run (lane = threadIdx.x, stride kBlock); 1: each wave owns U contiguous chunks (its 64 U pairs: 2 U KiB of f64 output,
// its operand loads of the U chunks adjacent) — tools/tune_store.hip "wave-contig", +0.6 % on the 3 B-read / 8 B-write mix.
#ifndef EC_WAVE_CONTIG
#define EC_WAVE_CONTIG 0
#endif
constexpr size_t tile_stride() { return EC_WAVE_CONTIG ? size_t(kWave) : size_t(kBlock); }
template <int U>
__device__ __forceinline__ size_t tile_lane_offset() {
    if constexpr (EC_WAVE_CONTIG) return size_t(threadIdx.x / kWave) * (size_t(kWave) * U) + (threadIdx.x & (kWave - 1));
    else return threadIdx.x;
}

// ---------------------------------------------------------------------------
// DIRECT variant: one block tile of kBlock*U pairs (2 cells each).  Pointers may sit
// at any cell offset (under-aligned accesses, ec_device.hpp); the cell-wise kernels run
// only when the "unaligned_vector" knob is off and a pointer is not 16-B aligned.
// ---------------------------------------------------------------------------
template <typename L, typename R, int OP, int U, bool NT_ST, bool NT_LD>
__device__ __forceinline__ void binop_direct_tile(const L* __restrict__ l, const R* __restrict__ r,
                                                  double* __restrict__ out, size_t npairs, size_t tile, unsigned cacheable) {
    using D2 = vec<double, 2>;
    constexpr bool FP = is_fp<L>::value || is_fp<R>::value;
    constexpr bool SM = is_small_int<L>::value && is_small_int<R>::value;  // 6-instruction exact divide (ec_device.hpp)
    constexpr size_t TILE = size_t(kBlock) * U;
    D2* __restrict__ op = reinterpret_cast<D2*>(out);
    constexpr size_t kStride = tile_stride();
    const size_t base = tile * TILE + tile_lane_offset<U>();
    if (tile * TILE + TILE <= npairs) {
        cells<L, 2> a[U];  // 1-byte operands travel as 16-bit words so that their loads keep `nt` (ec_device.hpp)
        cells<R, 2> b[U];
        policy_arms<2>(cacheable, [&](auto bits) {  // bit 0: l, bit 1: r loaded with the default cache policy (ec_device.hpp)
            constexpr unsigned B = decltype(bits)::value;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                a[j] = load_cells<NT_LD && !(B & 1u), L, 2>(l + 2 * (base + size_t(j) * kStride));
                b[j] = load_cells<NT_LD && !(B & 2u), R, 2>(r + 2 * (base + size_t(j) * kStride));
            }
        });
        if constexpr (OP == EC_DIV && SM && !FP) {
            // the short divide of small-integer cells with its zero-divisor case out of line: the tile's cells are tested once
            // (integer compares), and only a wave that holds a zero divisor runs the selects — 5 of the 11 vector instructions
            // per cell off the common path (with the u8 operand served from the Infinity Cache the divide is no longer fully
            // hidden behind memory: 0.4222 ms against the add's 0.4134 before this, profiles/r03/kernel_table.md)
            // Each quotient is a chain of six dependent FP64 instructions; written cell by cell the compiler issues the chains one
            // after the other, and with the u8 operand coming from HBM too the kernel is bound by how long a workgroup lives (its
            // occupancy is the hardware's maximum), so those ≈ 300 cycles show: the divide ran 2 % behind the add at equal bytes
            // (0.816 against 0.832, profiles/r04/store_policy_ab/).  Interleaved, rotating operand sets, three runs each: all 2 U
            // chains staged together 0.819-0.823; chunk by chunk 0.827 (the add: 0.8335).
#if EC_DIV_STAGE == 2
            // chunk by chunk — the chunk's two chains staged, its zero test, its store — so that the first store leaves as soon as
            // the first chunk's loads are back, while the second chunk's may still be in flight
#pragma unroll
            for (int j = 0; j < U; ++j) {
                double av[2], bv[2], q[2], y[2], e[2];
                bool zero = false;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    av[k] = to_f64(a[j][k]);
                    bv[k] = to_f64(b[j][k]);
                    zero = zero || b[j][k] == 0;
                }
                div_small_int_nonzero_staged<2>(av, bv, q, y, e);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(zero) != 0, 0)) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) q[i] = bv[i] == 0.0 ? div_by_zero(av[i]) : q[i];
                }
                store_vec<NT_ST>(op + base + size_t(j) * kStride, D2{q[0], q[1]});
            }
#else
            double av[2 * U], bv[2 * U], q[2 * U], y[2 * U], e[2 * U];
            bool zero = false;
#pragma unroll
            for (int j = 0; j < U; ++j)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    av[2 * j + k] = to_f64(a[j][k]);
                    bv[2 * j + k] = to_f64(b[j][k]);
                    zero = zero || b[j][k] == 0;
                }
            div_small_int_nonzero_staged<2 * U>(av, bv, q, y, e);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(zero) != 0, 0)) {
#pragma unroll
                for (int i = 0; i < 2 * U; ++i) q[i] = bv[i] == 0.0 ? div_by_zero(av[i]) : q[i];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) store_vec<NT_ST>(op + base + size_t(j) * kStride, D2{q[2 * j], q[2 * j + 1]});
#endif
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {  // chunk by chunk, the NaN rule tested once per chunk (cell_op_n, ec_device.hpp)
                const double av[2] = {to_f64(a[j][0]), to_f64(a[j][1])}, bv[2] = {to_f64(b[j][0]), to_f64(b[j][1])};
                double o[2];
                cell_op_n<OP, FP, SM, 2>(av, bv, o);
                store_vec<NT_ST>(op + base + size_t(j) * kStride, D2{o[0], o[1]});
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t p = base + size_t(j) * kStride;
            if (p < npairs) {
                const cells<L, 2> a = load_cells<NT_LD, L, 2>(l + 2 * p);
                const cells<R, 2> b = load_cells<NT_LD, R, 2>(r + 2 * p);
                D2 o;  // the ragged last tile: lanes diverge here anyway (p < npairs), the per-cell form is as good
                o.x = cell_op<OP, FP, SM>(to_f64(a[0]), to_f64(b[0]));
                o.y = cell_op<OP, FP, SM>(to_f64(a[1]), to_f64(b[1]));
                store_vec<NT_ST>(op + p, o);
            }
        }
    }
}

// One block per tile, straight-line: no grid-stride loop (profiles/r01/tune_binop_v2.log — the
// loop-free form runs ≈5 % faster than a grid capped at a few blocks per CU).  The grid must be
// exactly ceil(npairs / TILE) workgroups (two_front_tile is a permutation of the tile indices).
// `head` (0 or 1, chosen by the launcher: peel_head() in ec_runtime.hpp) leading cells are computed one by
// one by workgroup 0 and the pair grid starts after them, so that the 2-cell loads of 1-byte operands fall on
// even addresses (measurements and the rule: ec_runtime.hpp).  Bits 8.. of `head` carry the launch's load policy:
// bit 8 = l, bit 9 = r (masked kernels: bit 10 = lmask, bit 11 = rmask) is loaded cacheable instead of nt
// (cache_plan(), ec_runtime.hpp; policy_arms(), ec_device.hpp).
template <typename L, typename R, int OP, int U, bool NT_ST, bool NT_LD>
__device__ __forceinline__ void binop_direct_body(const L* __restrict__ l, const R* __restrict__ r,
                                                  double* __restrict__ out, size_t n, unsigned head_and_policy) {
    constexpr bool FP = is_fp<L>::value || is_fp<R>::value;
    constexpr bool SM = is_small_int<L>::value && is_small_int<R>::value;  // 6-instruction exact divide (ec_device.hpp)
    const unsigned head = head_and_policy & 0xffu, cacheable = head_and_policy >> 8;
    if (head) {
        if (blockIdx.x == 0 && threadIdx.x < head)
            st_cell(cell_op<OP, FP, SM>(to_f64(ld_cell(l + threadIdx.x)), to_f64(ld_cell(r + threadIdx.x))), out + threadIdx.x);
        l += head;
        r += head;
        out += head;
        n -= head;
    }
    binop_direct_tile<L, R, OP, U, NT_ST, NT_LD>(l, r, out, n >> 1, two_front_tile(), cacheable);
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        st_cell(cell_op<OP, FP, SM>(to_f64(ld_cell(l + n - 1)), to_f64(ld_cell(r + n - 1))), out + n - 1);
}

// NANRULE: cv_bin_op!'s NaN rule is compiled in.  The scalar may be any of the 10 types (widened to f64 on the host), so in general a result
// can be a NaN; but integer cells with a FINITE scalar (non-zero for a divide) cannot produce one — finite op finite is finite or ±inf — and
// the host then launches the form without the rule (u8 * 2.0, the reference's own example: 0.774 -> 0.80 with every byte from HBM).
template <typename L, int OP, int U, bool NT_ST, bool NT_LD, bool NANRULE = true>
__device__ __forceinline__ void binop_scalar_tile(const L* __restrict__ l, double s, double* __restrict__ out,
                                                  size_t npairs, size_t tile, unsigned cacheable) {
    using D2 = vec<double, 2>;
    constexpr bool FP = NANRULE;
    constexpr size_t TILE = size_t(kBlock) * U;
    D2* __restrict__ op = reinterpret_cast<D2*>(out);
    constexpr size_t kStride = tile_stride();
    const size_t base = tile * TILE + tile_lane_offset<U>();
    if (tile * TILE + TILE <= npairs) {
        cells<L, 2> a[U];
        policy_arms<1>(cacheable, [&](auto bits) {
            constexpr unsigned B = decltype(bits)::value;
#pragma unroll
            for (int j = 0; j < U; ++j) a[j] = load_cells<NT_LD && !(B & 1u), L, 2>(l + 2 * (base + size_t(j) * kStride));
        });
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const double av[2] = {to_f64(a[j][0]), to_f64(a[j][1])}, bv[2] = {s, s};
            double o[2];
            cell_op_n<OP, FP, false, 2>(av, bv, o);
            store_vec<NT_ST>(op + base + size_t(j) * kStride, D2{o[0], o[1]});
        }
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t p = base + size_t(j) * kStride;
            if (p < npairs) {
                const cells<L, 2> a = load_cells<NT_LD, L, 2>(l + 2 * p);
                D2 o;
                o.x = cell_op<OP, FP>(to_f64(a[0]), s);
                o.y = cell_op<OP, FP>(to_f64(a[1]), s);
                store_vec<NT_ST>(op + p, o);
            }
        }
    }
}

template <typename L, int OP, int U, bool NT_ST, bool NT_LD, bool NANRULE = true>
__global__ __launch_bounds__(kBlock) void k_binop_scalar_direct(const L* __restrict__ l, double s,
                                                                double* __restrict__ out, size_t n, unsigned head_and_policy) {
    const unsigned head = head_and_policy & 0xffu, cacheable = head_and_policy >> 8;
    if (head) {  // see binop_direct_body
        if (blockIdx.x == 0 && threadIdx.x < head) st_cell(cell_op<OP, true>(to_f64(ld_cell(l + threadIdx.x)), s), out + threadIdx.x);
        l += head;
        out += head;
        n -= head;
    }
    binop_scalar_tile<L, OP, U, NT_ST, NT_LD, NANRULE>(l, s, out, n >> 1, two_front_tile(), cacheable);
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) st_cell(cell_op<OP, true>(to_f64(ld_cell(l + n - 1)), s), out + n - 1);
}

// Any alignment, any n: one cell per lane. Correctness fallback for odd offsets.
template <typename L, typename R, int OP>
__global__ __launch_bounds__(kBlock) void k_binop_cellwise(const L* __restrict__ l, const R* __restrict__ r,
                                                           double* __restrict__ out, size_t n) {
    constexpr bool FP = is_fp<L>::value || is_fp<R>::value;
    constexpr bool SM = is_small_int<L>::value && is_small_int<R>::value;  // 6-instruction exact divide (ec_device.hpp)
    const size_t stride = size_t(gridDim.x) * kBlock;
    for (size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = cell_op<OP, FP, SM>(to_f64(l[i]), to_f64(r[i]));
}

template <typename L, int OP>
__global__ __launch_bounds__(kBlock) void k_binop_scalar_cellwise(const L* __restrict__ l, double s,
                                                                  double* __restrict__ out, size_t n) {
    const size_t stride = size_t(gridDim.x) * kBlock;
    for (size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = cell_op<OP, true>(to_f64(l[i]), s);
}

// ---------------------------------------------------------------------------
// LDS-staged variant ("LDS staging for the widening step").  Wave tile = 256 cells = two chunks of
// one dwordx4 store per lane (the tile depth that is fastest for DIRECT too).  An operand of ≤4
// bytes is fetched with ONE load of 4*sizeof(T) bytes per lane (dword / dwordx2 / dwordx4: half the
// global load instructions of DIRECT), written to a wave-private LDS slab, and each lane reads back
// the two cells that belong to each of its two 16-B output slots, widens them and stores.  The slab
// is private to the wave, so a wavefront-scope fence orders write -> read; no workgroup barrier.
// 8-byte operands already load 16 B per lane and go direct.
// ---------------------------------------------------------------------------
constexpr int kLdsChunks = 2;                        // chunks per wave tile
constexpr size_t kLdsWaveCells = 128 * kLdsChunks;   // 256 cells

template <typename T>
struct Staged {
    static constexpr bool value = sizeof(T) <= 4;
    using Load = vec<uint32_t, int(sizeof(T))>;                        // 4*sizeof(T) bytes per lane
    static constexpr int kSlabBytes = value ? int(kLdsWaveCells * sizeof(T)) : 16;
};

using u32x4 = vec<uint32_t, 4>;

template <typename L, typename R, int OP, bool NT_ST, bool NT_LD>
__device__ __forceinline__ void binop_lds_body(const L* __restrict__ l, const R* __restrict__ r,
                                               double* __restrict__ out, size_t n) {
    using L2 = vec<L, 2>;
    using R2 = vec<R, 2>;
    using D2 = vec<double, 2>;
    constexpr bool FP = is_fp<L>::value || is_fp<R>::value;
    constexpr bool SM = is_small_int<L>::value && is_small_int<R>::value;  // 6-instruction exact divide (ec_device.hpp)
    constexpr bool SL = Staged<L>::value, SR = Staged<R>::value;

    __shared__ __attribute__((aligned(16))) unsigned char slab_l[kWavesPerBlock][Staged<L>::kSlabBytes];
    __shared__ __attribute__((aligned(16))) unsigned char slab_r[kWavesPerBlock][Staged<R>::kSlabBytes];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const size_t nfull = n / kLdsWaveCells;  // whole wave tiles
    const size_t t = two_front_tile() * kWavesPerBlock + wave;
    if (t < nfull) {
        const size_t cell0 = t * kLdsWaveCells;
        L2 a[kLdsChunks];
        R2 b[kLdsChunks];
        typename Staged<L>::Load sl{};
        typename Staged<R>::Load sr{};
        if constexpr (SL) {
            sl = load_vec<NT_LD>(reinterpret_cast<const typename Staged<L>::Load*>(l + cell0) + lane);
        } else {
#pragma unroll
            for (int j = 0; j < kLdsChunks; ++j) a[j] = load_vec<NT_LD>(reinterpret_cast<const L2*>(l + cell0) + j * kWave + lane);
        }
        if constexpr (SR) {
            sr = load_vec<NT_LD>(reinterpret_cast<const typename Staged<R>::Load*>(r + cell0) + lane);
        } else {
#pragma unroll
            for (int j = 0; j < kLdsChunks; ++j) b[j] = load_vec<NT_LD>(reinterpret_cast<const R2*>(r + cell0) + j * kWave + lane);
        }
        if constexpr (SL) reinterpret_cast<typename Staged<L>::Load*>(slab_l[wave])[lane] = sl;
        if constexpr (SR) reinterpret_cast<typename Staged<R>::Load*>(slab_r[wave])[lane] = sr;
        if constexpr (SL || SR) {
            // the slab is private to this wave: order the wave's own LDS writes before its reads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if constexpr (SL) {
#pragma unroll
            for (int j = 0; j < kLdsChunks; ++j) a[j] = reinterpret_cast<const L2*>(slab_l[wave])[j * kWave + lane];
        }
        if constexpr (SR) {
#pragma unroll
            for (int j = 0; j < kLdsChunks; ++j) b[j] = reinterpret_cast<const R2*>(slab_r[wave])[j * kWave + lane];
        }
        D2* o2 = reinterpret_cast<D2*>(out + cell0);
#pragma unroll
        for (int j = 0; j < kLdsChunks; ++j) {
            const double av[2] = {to_f64(a[j].x), to_f64(a[j].y)}, bv[2] = {to_f64(b[j].x), to_f64(b[j].y)};
            double o[2];
            cell_op_n<OP, FP, SM, 2>(av, bv, o);
            store_vec<NT_ST>(o2 + j * kWave + lane, D2{o[0], o[1]});
        }
    }
    // ragged tail (< one wave tile): cell-wise by workgroup 0
    if (blockIdx.x == 0)
        for (size_t i = nfull * kLdsWaveCells + threadIdx.x; i < n; i += kBlock)
            st_cell(cell_op<OP, FP, SM>(to_f64(ld_cell(l + i)), to_f64(ld_cell(r + i))), out + i);
}

template <typename L, typename R, int OP, int U, bool NT_ST, bool NT_LD>
__global__ __launch_bounds__(kBlock) void k_binop_direct(const L* __restrict__ l, const R* __restrict__ r,
                                                         double* __restrict__ out, size_t n, unsigned head = 0) {
    binop_direct_body<L, R, OP, U, NT_ST, NT_LD>(l, r, out, n, head);
}

template <typename L, typename R, int OP, bool NT_ST, bool NT_LD>
__global__ __launch_bounds__(kBlock) void k_binop_lds(const L* __restrict__ l, const R* __restrict__ r,
                                                      double* __restrict__ out, size_t n) {
    binop_lds_body<L, R, OP, NT_ST, NT_LD>(l, r, out, n);
}

// `&Mask & &Mask` (src/masked/mask.rs:129-140) as a block-tiled 16-B-per-lane stream.
__device__ __forceinline__ void mask_and_body(const uint8_t* __restrict__ lm, const uint8_t* __restrict__ rm,
                                              uint8_t* __restrict__ om, size_t n, unsigned cacheable) {
    const size_t ngroups = n / 16;
    const u32x4* __restrict__ a = reinterpret_cast<const u32x4*>(lm);
    const u32x4* __restrict__ b = reinterpret_cast<const u32x4*>(rm);
    u32x4* __restrict__ o = reinterpret_cast<u32x4*>(om);
    const size_t stride = size_t(gridDim.x) * kBlock;
    for (size_t g = size_t(blockIdx.x) * kBlock + threadIdx.x; g < ngroups; g += stride) {
        u32x4 x, y;
        policy_arms<2>(cacheable >> 2, [&](auto bits) {  // bits 2, 3 of the launch's policy: a mask of a 16384² raster is 256 MiB
            constexpr unsigned B = decltype(bits)::value;
            x = load_vec<!(B & 1u)>(a + g);
            y = load_vec<!(B & 2u)>(b + g);
        });
        mask_store(x & y, o + g);
    }
    if (blockIdx.x == 0)
        for (size_t i = ngroups * 16 + threadIdx.x; i < n; i += kBlock) st_cell<uint8_t>(ld_cell(lm + i) & ld_cell(rm + i), om + i);
}

// impl $trt for &MaskedCellBuffer (src/masked/masked_buffer.rs:326-335) in one
// launch: the buffer op over ALL cells (masked-out cells are still computed,
// :331) and the mask AND (:333), each with its own lane->cell mapping so both
// streams stay 16 B per lane.
template <typename L, typename R, int OP, int U, bool NT_ST, bool NT_LD, bool LDS>
__global__ __launch_bounds__(kBlock) void k_masked_binop(const L* __restrict__ l, const uint8_t* __restrict__ lm,
                                                         const R* __restrict__ r, const uint8_t* __restrict__ rm,
                                                         double* __restrict__ out, uint8_t* __restrict__ om, size_t n,
                                                         unsigned head) {
    if constexpr (LDS) binop_lds_body<L, R, OP, NT_ST, NT_LD>(l, r, out, n);
    else binop_direct_body<L, R, OP, U, NT_ST, NT_LD>(l, r, out, n, head);
    mask_and_body(lm, rm, om, n, head >> 8);
}

template <typename L, typename R, int OP>
__global__ __launch_bounds__(kBlock) void k_masked_binop_cellwise(const L* __restrict__ l, const uint8_t* __restrict__ lm,
                                                                  const R* __restrict__ r, const uint8_t* __restrict__ rm,
                                                                  double* __restrict__ out, uint8_t* __restrict__ om, size_t n) {
    constexpr bool FP = is_fp<L>::value || is_fp<R>::value;
    constexpr bool SM = is_small_int<L>::value && is_small_int<R>::value;  // 6-instruction exact divide (ec_device.hpp)
    const size_t stride = size_t(gridDim.x) * kBlock;
    for (size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        out[i] = cell_op<OP, FP, SM>(to_f64(l[i]), to_f64(r[i]));
        om[i] = lm[i] & rm[i];
    }
}

}  // namespace ecd
