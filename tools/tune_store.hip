// tune_store.hip — the f64 STORE stream on its own (dev tool, not part of the library).
//
// Round 3 left every streaming kernel of the library at 0.78–0.83 of the 8 TB/s peak once all of its bytes come from
// and go to HBM, with the write stream the slowest part (write-only `fill<f64>` 0.81, read-only `min_max<f64>` 0.85) and
// the output 73 % of the headline's bytes.  This harness isolates that stream and sweeps what had not been isolated:
//
//   * contiguous bytes per wave per visit: 1, 2, 4, 8 KiB (U one-KiB `global_store_dwordx4` wave-instructions, either
//     wave-contiguous — a wave owns U KiB — or workgroup-interleaved — instruction j of all waves covers WAVES KiB, the
//     shipped shape with U = 2);
//   * workgroup -> address maps: linear, the shipped two fronts, each XCD inside its own contiguous eighth (single
//     front and two fronts per eighth), XCD-private chunks of 64 KiB / 2 MiB dealt round-robin, and persistent
//     workgroups (grid = CUs x k) walking grid-stride, their own contiguous range, or their XCD's eighth;
//   * resident waves per CU: workgroup size 64 … 1024 and an LDS reservation that caps the workgroups per CU;
//   * store cache policy: default, nt, sc1, sc0 sc1, nt sc1, nt sc0 sc1;
//
// for (a) the pure write of 8 B/cell (2 GiB at 16384²) and (b) the headline's byte mix, 3 B read + 8 B written per
// cell (u8 + u16 -> f64 with an add, nt loads, four operand sets in rotation so that no operand byte is served from the
// Infinity Cache).  Reference points: the runtime's own fill (`hipMemsetD32Async`) and a pure 16 B/lane read.
// Variants run in randomised order over interleaved rounds; every variant's output is checksummed (all must agree).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 [-DEC_STORE_SWEEP=1..8] -Iinclude -Ierased-cells_amd/csrc tools/tune_store.hip -o tools/tune_store
//   ./tools/tune_store [side=16384] [rounds=9] [iters=10] [only=<substring of a variant's name>]
// With `only` the program runs just the matching variants (no shuffling) — the form used directly after
// `rocprofv3 --pmc … --` for the write-credit-stall / request-level counters (tools/jobs/r04store.sh).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <utility>
#include <string>
#include <vector>

#include "ec_binop_kernels.hpp"

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

using namespace ecd;
using D2 = vec<double, 2>;

// store policies: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 nt sc0 sc1.  The asm forms end in `s_nop 1`: on gfx940+ a VALU write of
// the data registers of a store of more than 64 bits needs two wait states behind it, and the compiler's hazard recognizer does not
// look into inline asm (the first sweep ran without it: the pure-write variants' checksums differed, profiles/r04/tune_store_v1.log).
template <int POL>
__device__ __forceinline__ void store16(D2* p, D2 v) {
    if constexpr (POL == 0) *p = v;
    else if constexpr (POL == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// Workgroup -> tile maps.  `tiles` = number of workgroup tiles (host guarantees tiles % 16 == 0 for maps 2-4).
//   0 linear   1 two fronts (shipped)   2 XCD-contiguous eighths   3 XCD eighths, two fronts inside each
//   4 XCD-private chunks of CH tiles dealt round-robin (chunk c belongs to XCD c % 8)
template <int MAP, int CH>
__device__ __forceinline__ size_t map_tile(size_t b, size_t tiles) {
    if constexpr (MAP == 1) return (b & 1) ? tiles - 1 - (b >> 1) : (b >> 1);
    else if constexpr (MAP == 2) return (b & 7) * (tiles >> 3) + (b >> 3);
    else if constexpr (MAP == 3) {
        const size_t per = tiles >> 3, k = b >> 3;
        return (b & 7) * per + ((k & 1) ? per - 1 - (k >> 1) : (k >> 1));
    } else if constexpr (MAP == 4) {
        const size_t seq = b >> 3;  // this XCD's seq-th workgroup
        return ((seq / CH) * 8 + (b & 7)) * CH + seq % CH;
    } else return b;
}

// One workgroup tile = WAVES * U KiB of output.  WC: wave w owns U KiB contiguous; else instruction j of all waves covers
// WAVES KiB contiguous (the library's layout).  MIX: out = f64(u8) + f64(u16), pair loads (ushort / dword) nt; else a pure write.
template <int U, int WAVES, bool WC, int POL, bool MIX, bool NTL = true, bool WAIT = false>
__device__ __forceinline__ void store_tile(const uint8_t* __restrict__ l, const uint16_t* __restrict__ r, D2* __restrict__ op, size_t tile) {
    constexpr int BLOCK = WAVES * 64;
    size_t base, stride;
    if constexpr (WC) {
        base = tile * (size_t(BLOCK) * U) + size_t(threadIdx.x >> 6) * (64 * U) + (threadIdx.x & 63);
        stride = 64;
    } else {
        base = tile * (size_t(BLOCK) * U) + threadIdx.x;
        stride = BLOCK;
    }
    if constexpr (MIX) {
        cells<uint8_t, 2> a[U];
        cells<uint16_t, 2> b[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            a[j] = load_cells<NTL, uint8_t, 2>(l + 2 * (base + j * stride));
            b[j] = load_cells<NTL, uint16_t, 2>(r + 2 * (base + j * stride));
        }
#pragma unroll
        for (int j = 0; j < U; ++j)
            store16<POL>(op + base + j * stride, D2{to_f64(a[j][0]) + to_f64(b[j][0]), to_f64(a[j][1]) + to_f64(b[j][1])});
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t p = base + j * stride;
            store16<POL>(op + p, D2{double(p), 1.0});
        }
    }
    // WAIT: the wave stays until its stores have been acknowledged, so the stores in flight are bounded by the resident waves
    if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int LDSKB>
__device__ __forceinline__ void reserve_lds() {
    if constexpr (LDSKB > 0) {  // occupancy cap: fewer workgroups fit per CU (160 KiB of LDS)
        __shared__ volatile uint32_t pad[LDSKB * 256];
        if (threadIdx.x == 0) pad[0] = 1;
    }
}

// one workgroup per tile
template <int U, int WAVES, bool WC, int POL, int MAP, int CH, int LDSKB, bool MIX, bool NTL = true, bool WAIT = false>
__global__ __launch_bounds__(WAVES * 64) void k_store(const uint8_t* __restrict__ l, const uint16_t* __restrict__ r,
                                                      D2* __restrict__ op, size_t tiles) {
    reserve_lds<LDSKB>();
    store_tile<U, WAVES, WC, POL, MIX, NTL, WAIT>(l, r, op, map_tile<MAP, CH>(blockIdx.x, tiles));
}

// The mix with WIDE loads: a wave fetches its whole tile of 128 U cells of each operand with ONE load per lane (u8: 2 U bytes per lane,
// u16: 4 U bytes per lane — U = 2: dword / dwordx2, U = 4: dwordx2 / dwordx4), parks them in a wave-private LDS slab and reads back
// the pair of cells of each 16-byte output slot (the library's k_binop_lds shape).  Half (U = 2) or a quarter (U = 4) of the load
// instructions of the direct form, each moving 2-4x the bytes: the same bytes in flight from fewer resident waves — the question
// being whether the store stream then likes the lower occupancy as much as the pure write does.
template <int U, int WAVES, int POL, int LDSKB>
__global__ __launch_bounds__(WAVES * 64) void k_mix_wide(const uint8_t* __restrict__ l, const uint16_t* __restrict__ r, D2* __restrict__ op, size_t tiles) {
    reserve_lds<LDSKB>();
    using LA = vec<uint32_t, U / 2>;  // 2 U bytes per lane
    using LB = vec<uint32_t, U>;      // 4 U bytes per lane
    __shared__ __attribute__((aligned(16))) unsigned char slab_a[WAVES][128 * U];
    __shared__ __attribute__((aligned(16))) unsigned char slab_b[WAVES][256 * U];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t b = blockIdx.x, tile = (b & 1) ? tiles - 1 - (b >> 1) : (b >> 1);
    const size_t cell0 = (tile * WAVES + wave) * (128 * U);
    const LA av = nt_load(reinterpret_cast<const LA*>(l + cell0) + lane);
    const LB bv = nt_load(reinterpret_cast<const LB*>(r + cell0) + lane);
    reinterpret_cast<LA*>(slab_a[wave])[lane] = av;
    reinterpret_cast<LB*>(slab_b[wave])[lane] = bv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    D2* o2 = op + cell0 / 2;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint16_t a2 = reinterpret_cast<const uint16_t*>(slab_a[wave])[j * 64 + lane];
        const uint32_t b2 = reinterpret_cast<const uint32_t*>(slab_b[wave])[j * 64 + lane];
        store16<POL>(o2 + j * 64 + lane, D2{double(a2 & 0xffu) + double(b2 & 0xffffu), double(a2 >> 8) + double(b2 >> 16)});
    }
}

// Seventh sweep: buffer * scalar over 1-byte cells (1 B read + 8 B written per cell: the library's most write-heavy common kernel, 0.79)
// with WIDE loads under occupancy caps.  A wave fetches its 128 U cells with one load of 2 U bytes per lane (U = 2: dword, 4: dwordx2,
// 8: dwordx4), parks them in a wave-private LDS slab and reads back the pair of cells of each 16-byte output slot: the operand bytes in
// flight come from 1/U of the load instructions, so the kernel may tolerate the few resident workgroups the store stream likes (fill: 0.95).
template <int U, int WAVES, int POL, int LDSKB>
__global__ __launch_bounds__(WAVES * 64) void k_sca_wide(const uint8_t* __restrict__ l, D2* __restrict__ op, size_t tiles) {
    reserve_lds<LDSKB>();
    using LA = vec<uint32_t, U / 2>;  // 2 U bytes per lane
    __shared__ __attribute__((aligned(16))) unsigned char slab[WAVES][128 * U];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t b = blockIdx.x, tile = (b & 1) ? tiles - 1 - (b >> 1) : (b >> 1);
    const size_t cell0 = (tile * WAVES + wave) * (128 * U);
    const LA av = nt_load(reinterpret_cast<const LA*>(l + cell0) + lane);
    reinterpret_cast<LA*>(slab[wave])[lane] = av;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    D2* o2 = op + cell0 / 2;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint16_t a2 = reinterpret_cast<const uint16_t*>(slab[wave])[j * 64 + lane];
        store16<POL>(o2 + j * 64 + lane, D2{double(a2 & 0xffu) * 2.0, double(a2 >> 8) * 2.0});
    }
}

// Eighth sweep: the mix with 16 bytes per lane of BOTH operands in flight per wave — a wave tile of 1024 cells: one dwordx4 of the u8 operand,
// two of the u16 operand, 8 KiB of output in eight stores — so that the few resident waves the store stream likes (fill: 2 workgroups per CU,
// 0.95) still carry the ~12 KB per CU of loads the read stream needs.  Stores without a "memory" clobber, as the library's (ec_device.hpp).
__device__ __forceinline__ void store16_free(D2* p, D2 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v)); }
template <int WAVES, int LDSKB, bool CLOBBER>
__global__ __launch_bounds__(WAVES * 64) void k_mix_wide8(const uint8_t* __restrict__ l, const uint16_t* __restrict__ r, D2* __restrict__ op, size_t tiles) {
    reserve_lds<LDSKB>();
    __shared__ __attribute__((aligned(16))) unsigned char slab_a[WAVES][1024];
    __shared__ __attribute__((aligned(16))) unsigned char slab_b[WAVES][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t b = blockIdx.x, tile = (b & 1) ? tiles - 1 - (b >> 1) : (b >> 1);
    const size_t cell0 = (tile * WAVES + wave) * 1024;
    const u32x4 av = nt_load(reinterpret_cast<const u32x4*>(l + cell0) + lane);
    const u32x4 bv0 = nt_load(reinterpret_cast<const u32x4*>(r + cell0) + lane);
    const u32x4 bv1 = nt_load(reinterpret_cast<const u32x4*>(r + cell0) + 64 + lane);
    reinterpret_cast<u32x4*>(slab_a[wave])[lane] = av;
    reinterpret_cast<u32x4*>(slab_b[wave])[lane] = bv0;
    reinterpret_cast<u32x4*>(slab_b[wave])[64 + lane] = bv1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    D2* o2 = op + cell0 / 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint16_t a2 = reinterpret_cast<const uint16_t*>(slab_a[wave])[j * 64 + lane];
        const uint32_t b2 = reinterpret_cast<const uint32_t*>(slab_b[wave])[j * 64 + lane];
        const D2 o = D2{double(a2 & 0xffu) + double(b2 & 0xffffu), double(a2 >> 8) + double(b2 >> 16)};
        if constexpr (CLOBBER) store16<4>(o2 + j * 64 + lane, o);
        else store16_free(o2 + j * 64 + lane, o);
    }
}

// Workgroup-wide loads: ONE wave-instruction fetches a whole KiB of a narrow operand for the workgroup (lane-contiguous 16 B per lane), the
// cells reach the four waves through LDS behind a workgroup barrier.  The direct form reads the same KiB as eight 128-byte requests from four
// waves; the question is whether the DRAM interface likes a write stream interrupted by a few long reads better than by many short ones.
// SCALAR: out = f64(u8) * 2 (1 B read + 8 B written per cell, the library's worst common kernel: 0.75); else the u8 + u16 mix.
template <int POL, bool SCALAR, bool WIDE>
__global__ __launch_bounds__(256) void k_wgwide(const uint8_t* __restrict__ l, const uint16_t* __restrict__ r, D2* __restrict__ op, size_t tiles) {
    __shared__ __attribute__((aligned(16))) unsigned char sa[1024];
    __shared__ __attribute__((aligned(16))) unsigned char sb[2048];
    const size_t b = blockIdx.x, tile = (b & 1) ? tiles - 1 - (b >> 1) : (b >> 1);
    const size_t cell0 = tile * 1024;  // 1024 cells per workgroup: 4 waves x 2 chunks of 128 cells
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t a2[2];
    uint32_t b2[2] = {0, 0};
    if constexpr (WIDE) {
        if (wave == 0) reinterpret_cast<u32x4*>(sa)[lane] = nt_load(reinterpret_cast<const u32x4*>(l + cell0) + lane);
        if constexpr (!SCALAR) {
            if (wave == 1) reinterpret_cast<u32x4*>(sb)[lane] = nt_load(reinterpret_cast<const u32x4*>(r + cell0) + lane);
            if (wave == 2) reinterpret_cast<u32x4*>(sb)[64 + lane] = nt_load(reinterpret_cast<const u32x4*>(r + cell0) + 64 + lane);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pr = j * 256 + threadIdx.x;  // pair index inside the tile, the library's workgroup-interleaved layout
            a2[j] = reinterpret_cast<const uint16_t*>(sa)[pr];
            if constexpr (!SCALAR) b2[j] = reinterpret_cast<const uint32_t*>(sb)[pr];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t pr = cell0 / 2 + j * 256 + threadIdx.x;
            a2[j] = nt_load(reinterpret_cast<const uint16_t*>(l) + pr);
            if constexpr (!SCALAR) b2[j] = nt_load(reinterpret_cast<const uint32_t*>(r) + pr);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const size_t pr = cell0 / 2 + j * 256 + threadIdx.x;
        D2 o;
        if constexpr (SCALAR) o = D2{double(a2[j] & 0xffu) * 2.0, double(a2[j] >> 8) * 2.0};
        else o = D2{double(a2[j] & 0xffu) + double(b2[j] & 0xffffu), double(a2[j] >> 8) + double(b2[j] >> 16)};
        store16<POL>(op + pr, o);
    }
}

// the library's buffer ∘ scalar tile (k_binop_scalar_direct, full tiles only) with the NaN rule of cv_bin_op! switchable
template <bool NANRULE>
__global__ __launch_bounds__(256) void k_scalar_lib_shape(const uint8_t* __restrict__ l, double sc, double* __restrict__ out, size_t n) {
    D2* __restrict__ op = reinterpret_cast<D2*>(out);
    const size_t base = two_front_tile() * 512 + threadIdx.x;
    cells<uint8_t, 2> a[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) a[j] = load_cells<true, uint8_t, 2>(l + 2 * (base + size_t(j) * 256));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        D2 o;
        o.x = cell_op<EC_MUL, NANRULE>(to_f64(a[j][0]), sc);
        o.y = cell_op<EC_MUL, NANRULE>(to_f64(a[j][1]), sc);
        nt_store(o, op + base + size_t(j) * 256);
    }
}

// Where a wave's life goes: the shipped mix tile with four timestamps per wave (s_memrealtime, 100 MHz) — start, operands back (first use),
// stores issued, stores acknowledged (s_waitcnt vmcnt(0)) — written for one wave in 64.  The kernel is bound by how long a workgroup lives
// (its occupancy is the hardware's maximum): this says which part of that life is load latency and which is the wait for the stores.
template <int POL>
__global__ __launch_bounds__(256) void k_mix_timed(const uint8_t* __restrict__ l, const uint16_t* __restrict__ r, D2* __restrict__ op, size_t tiles,
                                                   unsigned* __restrict__ stats) {
    const unsigned long long t0 = wall_clock64();
    const size_t b = blockIdx.x, tile = (b & 1) ? tiles - 1 - (b >> 1) : (b >> 1);
    const size_t base = tile * 512 + threadIdx.x;
    cells<uint8_t, 2> a[2];
    cells<uint16_t, 2> c[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        a[j] = load_cells<true, uint8_t, 2>(l + 2 * (base + j * 256));
        c[j] = load_cells<true, uint16_t, 2>(r + 2 * (base + j * 256));
    }
    D2 o[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) o[j] = D2{to_f64(a[j][0]) + to_f64(c[j][0]), to_f64(a[j][1]) + to_f64(c[j][1])};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = wall_clock64();
#pragma unroll
    for (int j = 0; j < 2; ++j) store16<POL>(op + base + j * 256, o[j]);
    const unsigned long long t2 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = wall_clock64();
    if ((b & 63) == 0 && (threadIdx.x & 63) == 0) {
        unsigned* q = stats + ((b >> 6) * 4 + (threadIdx.x >> 6)) * 3;
        q[0] = unsigned(t1 - t0);
        q[1] = unsigned(t2 - t1);
        q[2] = unsigned(t3 - t2);
    }
}

// persistent workgroups.  WALK 0: grid-stride (tile = b + k * grid)   1: workgroup b walks its own contiguous range
//                         2: the workgroups of an XCD (b % 8) walk that XCD's eighth, stride = grid / 8
template <int U, int WAVES, bool WC, int POL, int WALK, bool MIX>
__global__ __launch_bounds__(WAVES * 64) void k_store_persistent(const uint8_t* __restrict__ l, const uint16_t* __restrict__ r,
                                                                 D2* __restrict__ op, size_t tiles) {
    const size_t g = gridDim.x, b = blockIdx.x;
    if constexpr (WALK == 0) {
        for (size_t t = b; t < tiles; t += g) store_tile<U, WAVES, WC, POL, MIX>(l, r, op, t);
    } else if constexpr (WALK == 1) {
        const size_t per = (tiles + g - 1) / g, lo = b * per, hi = lo + per < tiles ? lo + per : tiles;
        for (size_t t = lo; t < hi; ++t) store_tile<U, WAVES, WC, POL, MIX>(l, r, op, t);
    } else {
        const size_t per = tiles >> 3, lo = (b & 7) * per, gx = g >> 3;
        for (size_t t = b >> 3; t < per; t += gx) store_tile<U, WAVES, WC, POL, MIX>(l, r, op, lo + t);
    }
}

// Fifth sweep: PERSISTENT workgroups with the operand loads SOFTWARE-PIPELINED.  The one-tile-per-workgroup kernel is bound by how long a
// workgroup lives (wave-life: 1.4 us waiting for operands, 1.0 us waiting for the store acknowledgement, no load in flight during the
// latter); the persistent variants of sweeps 1-2 walked their tiles one after the other (load, wait, store, load ...) and lost.  Here a
// wave keeps D tiles of operands in flight: stage s is computed and stored, then reloaded with the tile D steps ahead, while the other
// D - 1 stages' loads are still outstanding.  The stores are buffer stores through the compiler's own builtin (aux = sc1 | nt), so the
// compiler's s_waitcnt insertion sees them and waits with vmcnt(N > 0) — behind inline-asm stores it cannot count them and waits for
// the store acknowledgement too.  ASSIGN 0: tile = b + k * grid   1: the same through the two-fronts map   2: next tile from an atomic
// counter (one fetch-add per workgroup and tile, broadcast through LDS): the window of tiles in work stays as tight as the dispatcher
// keeps it for one-tile workgroups, whatever the drift between workgroups.
__global__ void k_set(unsigned* p, unsigned v) { *p = v; }
template <class F, int... S>
__device__ __forceinline__ bool any_stage(F& f, std::integer_sequence<int, S...>) {
    return (f(std::integral_constant<int, S>{}) || ...);
}
// ASSIGN 0: stage s of step k works on tile b + (k D + s) grid   1: the same through the two-fronts map
//        3: CHUNKS of D adjacent tiles: step k works on chunk b + k grid, stage s on its s-th tile
//        2: chunks from an atomic counter, one fetch-add per workgroup and chunk, issued inside stage 0 BEFORE that stage's loads and taken
//           one step later behind the wait for those loads (vmcnt counts in order: it has returned by then, at no extra wait).
// Everything that touches memory is inline asm with hand-placed s_waitcnt: the compiler, left to itself, sank the prefetches to their
// uses (restrict + const: no memory clobber holds them), hoisted the next stage's arithmetic above them, and merged the waits of the
// prologue and the loop into vmcnt(3) (three builds' ISA, all without a pipeline).  The loaded registers reach their uses only through
// the "+v" operands of the asm statement behind the wait.
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int U, int D, int ASSIGN>
__global__ __launch_bounds__(256) void k_mix_pipe(const uint8_t* __restrict__ l, const uint16_t* __restrict__ r, D2* __restrict__ op, size_t tiles,
                                                  unsigned* __restrict__ counter) {
    const size_t g = gridDim.x;
    constexpr bool CHUNKS = ASSIGN >= 2;
    static_assert((D - 1) * 3 * U + 1 <= 63, "vmcnt is a 6-bit counter");
    static_assert(D * U <= 8, "more stages: the stage arrays go to scratch and the asm loads' results are copied there before they have arrived (tools/inflight_check.py)");
    __shared__ unsigned handoff[2];
    uint32_t a[D][U], c[D][U];
    size_t cur = blockIdx.x, nxt = CHUNKS ? blockIdx.x + g : blockIdx.x + D * g;  // chunk (or step) of the tiles being stored / being loaded
    unsigned step = 0, fut = 0;
    const unsigned one = 1;
    auto tile_of = [&](size_t q, int s) -> size_t {
        if constexpr (CHUNKS) return q * D + s;
        else return q + s * g;  // q = b + k D g
    };
    auto where = [&](size_t v) -> size_t {
        if constexpr (ASSIGN == 1) return (v & 1) ? tiles - 1 - (v >> 1) : (v >> 1);
        else return v;
    };
    auto load = [&](int s, size_t q) {  // unconditional: past the end the last tile again, unused
        const size_t t = tile_of(q, s);
        const size_t base = where(t < tiles ? t : tiles - 1) * (256 * U) + threadIdx.x;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            asm volatile("global_load_ushort %0, %1, off nt" : "=v"(a[s][j]) : "v"(reinterpret_cast<const uint16_t*>(l) + base + j * 256) : "memory");
            asm volatile("global_load_dword %0, %1, off nt" : "=v"(c[s][j]) : "v"(reinterpret_cast<const uint32_t*>(r) + base + j * 256) : "memory");
        }
    };
    auto fetch = [&]() {
        if (threadIdx.x == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(fut) : "v"(counter), "v"(one) : "memory");
    };
    if constexpr (ASSIGN == 2) {  // the counter starts at grid: chunk b is this workgroup's first, the next comes from the queue
        fetch();
        wait_vm<0>();
        asm volatile("" : "+v"(fut));
        if (threadIdx.x == 0) handoff[1] = fut;
        __syncthreads();
        nxt = handoff[1];
        fetch();  // for step 1, taken in step 0's stage 0... no: taken in step 1's stage 0
    }
#pragma unroll
    for (int s = 0; s < D; ++s) load(s, cur);
    // FIRST: the pass over the prologue's loads — behind the loads of stage s there are the loads of the later stages (2 U each) and what the
    // earlier stages of this pass issued (U stores + 2 U loads each); afterwards always the other D - 1 stages' stores and loads.
    auto stage = [&](auto S, auto First) -> bool {  // true: this workgroup has run out of tiles
        constexpr int s = decltype(S)::value;
        constexpr bool FIRST = decltype(First)::value;
        const size_t t = tile_of(cur, s);
        if (t >= tiles) return true;
        wait_vm<FIRST ? (D - 1 - s) * 2 * U + s * 3 * U : (D - 1) * 3 * U>();
#pragma unroll
        for (int j = 0; j < U; ++j) asm volatile("" : "+v"(a[s][j]), "+v"(c[s][j]));
        if constexpr (ASSIGN == 2 && s == 0 && !FIRST) {
            asm volatile("" : "+v"(fut));
            if (threadIdx.x == 0) handoff[step & 1] = fut;
            fetch();
            __syncthreads();
            nxt = handoff[step & 1];
        }
        const size_t base = where(t) * (256 * U) + threadIdx.x;
#pragma unroll
        for (int j = 0; j < U; ++j)
            store16<4>(op + base + j * 256, D2{double(a[s][j] & 0xffu) + double(c[s][j] & 0xffffu), double((a[s][j] >> 8) & 0xffu) + double(c[s][j] >> 16)});
        load(s, nxt);
        return false;
    };
    auto pass = [&](auto First) -> bool {
        return [&]<int... S>(std::integer_sequence<int, S...>) { return (stage(std::integral_constant<int, S>{}, First) || ...); }
        (std::make_integer_sequence<int, D>{});
    };
    bool done = pass(std::true_type{});
    while (!done) {
        cur = nxt;
        ++step;
        if constexpr (ASSIGN != 2) nxt += CHUNKS ? g : D * g;
        done = pass(std::false_type{});
    }
    wait_vm<0>();
}

__global__ void k_fill(uint8_t* a, uint16_t* b, size_t n, uint64_t seed) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        a[i] = uint8_t(splitmix64((seed + 1) ^ i));
        b[i] = uint16_t(splitmix64((seed + 2) ^ i) | 1);
    }
}
__global__ void k_checksum(const uint64_t* p, size_t n, unsigned long long* acc) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    unsigned long long s = 0;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) s += p[i] * (i | 1);
    atomicAdd(acc, s);
}
template <int U>
__global__ __launch_bounds__(256) void k_read_tile(const u32x4* __restrict__ s, uint32_t* sink) {
    const size_t b = blockIdx.x, tile = (b & 1) ? gridDim.x - 1 - (b >> 1) : (b >> 1);
    size_t base = tile * 256 * U + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < U; ++j) acc ^= nt_load(s + base + j * 256);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) *sink = 1;
}

// read-only shapes over a SMALL stream (n bytes: the 1 B/cell reductions' 256 MiB at 16384²), to see what a 40 µs kernel can reach at all
template <int BLOCK, int U, bool FRONTS>
__global__ __launch_bounds__(BLOCK) void k_read_small_tile(const u32x4* __restrict__ s, uint32_t* sink) {
    const size_t b = blockIdx.x, tile = FRONTS ? ((b & 1) ? gridDim.x - 1 - (b >> 1) : (b >> 1)) : b;
    const size_t base = tile * BLOCK * U + threadIdx.x;
    u32x4 x[U];
#pragma unroll
    for (int j = 0; j < U; ++j) x[j] = nt_load(s + base + j * BLOCK);
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < U; ++j) acc ^= x[j];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) *sink = 1;
}
template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void k_read_small_persistent(const u32x4* __restrict__ s, size_t tiles, uint32_t* sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t base = t * BLOCK * U + threadIdx.x;
        u32x4 x[U];
#pragma unroll
        for (int j = 0; j < U; ++j) x[j] = nt_load(s + base + j * BLOCK);
#pragma unroll
        for (int j = 0; j < U; ++j) acc ^= x[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) *sink = 1;
}

struct Variant {
    std::string name;
    std::function<void(int)> launch;  // argument: the launch's number (MIX rotates its operand set with it)
    double bytes;
    bool mix;
    std::vector<float> ms;
};

int main(int argc, char** argv) {
    const size_t side = argc > 1 ? strtoull(argv[1], 0, 10) : 16384;
    const int rounds = argc > 2 ? atoi(argv[2]) : 9;
    const int iters = argc > 3 ? atoi(argv[3]) : 10;
    const char* only = argc > 4 ? argv[4] : nullptr;
    const size_t n = side * side, npairs = n / 2;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  n %zu cells (output %.2f GiB)\n", prop.gcnArchName, cus, n, double(n) * 8 / (1 << 30));
    constexpr int SETS = 4;
    uint8_t* a[SETS];
    uint16_t* b[SETS];
    double* out[2];
    unsigned long long* acc;
    uint32_t* sink;
    for (int k = 0; k < SETS; ++k) {
        CK(hipMalloc(&a[k], n));
        CK(hipMalloc(&b[k], n * 2));
        k_fill<<<2048, 256>>>(a[k], b[k], n, 0x5EED0000 + 16 * k);
    }
    // argv[5]: how the OUTPUT is allocated — "uncached" (hipDeviceMallocUncached: MTYPE UC, stores and loads bypass the L2s) or
    // "finegrained" (hipDeviceMallocFinegrained) instead of plain hipMalloc: the library hands out its buffers itself (ec_alloc)
    const char* out_alloc = argc > 5 ? argv[5] : "default";
    for (int k = 0; k < 2; ++k) {
        if (!strcmp(out_alloc, "uncached")) CK(hipExtMallocWithFlags((void**)&out[k], n * 8, hipDeviceMallocUncached));
        else if (!strcmp(out_alloc, "finegrained")) CK(hipExtMallocWithFlags((void**)&out[k], n * 8, hipDeviceMallocFinegrained));
        else CK(hipMalloc(&out[k], n * 8));
    }
    printf("output allocation: %s\n", out_alloc);
    CK(hipMalloc(&acc, 8));
    CK(hipMalloc(&sink, 4));
    CK(hipDeviceSynchronize());

    std::vector<Variant> vs;
    auto add = [&](std::string name, bool mix, std::function<void(int)> f) {
        if (only && name.find(only) == std::string::npos) return;
        vs.push_back(Variant{name, f, (mix ? 11.0 : 8.0) * double(n), mix, {}});
    };
    const char* polname[] = {"plain", "nt", "sc1", "sc0sc1", "nt+sc1", "nt+sc0sc1"};
    const char* mapname[] = {"linear", "2fronts", "xcd-eighths", "xcd-eighths-2fronts", "xcd-chunks"};

// one workgroup per tile: U, WAVES, WC, POL, MAP, CH (tiles per XCD chunk, map 4), LDSKB, MIX
#define ST(U, WAVES, WC, POL, MAP, CH, LDSKB, MIX) STL(U, WAVES, WC, POL, MAP, CH, LDSKB, MIX, true)
#define STL(U, WAVES, WC, POL, MAP, CH, LDSKB, MIX, NTL)                                                                               \
    if (npairs % (size_t(WAVES) * 64 * U) == 0 && (npairs / (size_t(WAVES) * 64 * U)) % (8 * (CH)) == 0) {                            \
        char nm[160];                                                                                                                  \
        snprintf(nm, sizeof nm, "%s U%d x%dw %s %s %s%s lds%dK%s", MIX ? "mix" : "wr ", U, WAVES, WC ? "wave-contig" : "wg-interleave", \
                 polname[POL], mapname[MAP], MAP == 4 ? (" ch" #CH) : "", LDSKB, NTL ? "" : " plain-loads");                           \
        const size_t tiles = npairs / (size_t(WAVES) * 64 * U);                                                                        \
        add(nm, MIX, [=](int i) {                                                                                                      \
            k_store<U, WAVES, WC, POL, MAP, CH, LDSKB, MIX, NTL><<<unsigned(tiles), WAVES * 64>>>(a[i % SETS], b[i % SETS], (D2*)out[i & 1], tiles); \
        });                                                                                                                            \
    }
// persistent: U, WAVES, WC, POL, WALK, workgroups per CU, MIX
#define PS(U, WAVES, WC, POL, WALK, BPC, MIX)                                                                                          \
    if (npairs % (size_t(WAVES) * 64 * U) == 0 && (npairs / (size_t(WAVES) * 64 * U)) % 8 == 0) {                                     \
        char nm[160];                                                                                                                  \
        snprintf(nm, sizeof nm, "%s U%d x%dw %s %s persistent %s %d/CU", MIX ? "mix" : "wr ", U, WAVES, WC ? "wave-contig" : "wg-interleave", \
                 polname[POL], WALK == 0 ? "grid-stride" : WALK == 1 ? "own-range" : "xcd-eighth", BPC);                              \
        const size_t tiles = npairs / (size_t(WAVES) * 64 * U);                                                                        \
        add(nm, MIX, [=](int i) {                                                                                                      \
            k_store_persistent<U, WAVES, WC, POL, WALK, MIX><<<unsigned(cus * BPC), WAVES * 64>>>(a[i % SETS], b[i % SETS], (D2*)out[i & 1], tiles); \
        });                                                                                                                            \
    }

#define SWEEP1(MIX)                                                                                    \
    /* the shipped shape first: U2, 4 waves, workgroup-interleaved, nt, two fronts */                  \
    ST(2, 4, false, 1, 1, 1, 0, MIX)                                                                   \
    /* contiguous bytes per wave per visit */                                                          \
    ST(1, 4, false, 1, 1, 1, 0, MIX)                                                                   \
    ST(2, 4, true, 1, 1, 1, 0, MIX)                                                                    \
    ST(4, 4, false, 1, 1, 1, 0, MIX)                                                                   \
    ST(4, 4, true, 1, 1, 1, 0, MIX)                                                                    \
    ST(8, 4, false, 1, 1, 1, 0, MIX)                                                                   \
    ST(8, 4, true, 1, 1, 1, 0, MIX)                                                                    \
    /* workgroup -> address maps */                                                                    \
    ST(2, 4, false, 1, 0, 1, 0, MIX)                                                                   \
    ST(2, 4, false, 1, 2, 1, 0, MIX)                                                                   \
    ST(2, 4, false, 1, 3, 1, 0, MIX)                                                                   \
    ST(2, 4, false, 1, 4, 8, 0, MIX)                                                                   \
    ST(2, 4, false, 1, 4, 256, 0, MIX)                                                                 \
    ST(4, 4, true, 1, 2, 1, 0, MIX)                                                                    \
    ST(4, 4, true, 1, 3, 1, 0, MIX)                                                                    \
    ST(8, 4, true, 1, 3, 1, 0, MIX)                                                                    \
    /* waves per workgroup */                                                                          \
    ST(2, 1, false, 1, 1, 1, 0, MIX)                                                                   \
    ST(2, 2, false, 1, 1, 1, 0, MIX)                                                                   \
    ST(2, 8, false, 1, 1, 1, 0, MIX)                                                                   \
    ST(2, 16, false, 1, 1, 1, 0, MIX)                                                                  \
    ST(8, 1, true, 1, 1, 1, 0, MIX)                                                                    \
    /* resident workgroups per CU capped by an LDS reservation: 64 K -> 2, 32 K -> 5, 16 K -> 10 (of 8 by waves) */ \
    ST(2, 4, false, 1, 1, 1, 64, MIX)                                                                  \
    ST(2, 4, false, 1, 1, 1, 32, MIX)                                                                  \
    ST(8, 4, true, 1, 1, 1, 64, MIX)                                                                   \
    ST(8, 4, true, 1, 1, 1, 32, MIX)                                                                   \
    /* store policy */                                                                                 \
    ST(2, 4, false, 0, 1, 1, 0, MIX)                                                                   \
    ST(2, 4, false, 2, 1, 1, 0, MIX)                                                                   \
    ST(2, 4, false, 3, 1, 1, 0, MIX)                                                                   \
    ST(2, 4, false, 4, 1, 1, 0, MIX)                                                                   \
    ST(2, 4, false, 5, 1, 1, 0, MIX)                                                                   \
    ST(2, 4, false, 0, 3, 1, 0, MIX)                                                                   \
    ST(2, 4, false, 2, 3, 1, 0, MIX)                                                                   \
    /* persistent workgroups */                                                                        \
    PS(2, 4, false, 1, 0, 8, MIX)                                                                      \
    PS(2, 4, false, 1, 0, 4, MIX)                                                                      \
    PS(4, 4, true, 1, 1, 8, MIX)                                                                       \
    PS(4, 4, true, 1, 1, 2, MIX)                                                                       \
    PS(2, 4, false, 1, 2, 8, MIX)                                                                      \
    PS(4, 4, true, 1, 2, 4, MIX)                                                                       \
    PS(8, 4, true, 1, 2, 2, MIX)

// second sweep (after profiles/r04/tune_store_v1.log): what the first one pointed at — fewer resident workgroups for the pure
// write, write-through stores (sc1) for the mix — crossed with each other and with the tile shape
#define SWEEP2_WRITE                                                                                   \
    ST(2, 4, false, 1, 1, 1, 0, false)                                                                 \
    ST(2, 4, false, 1, 1, 1, 16, false)                                                                \
    ST(2, 4, false, 1, 1, 1, 24, false)                                                                \
    ST(2, 4, false, 1, 1, 1, 40, false)                                                                \
    ST(2, 4, false, 1, 1, 1, 48, false)                                                                \
    ST(2, 4, false, 1, 1, 1, 64, false)                                                                \
    ST(2, 4, false, 1, 1, 1, 96, false)                                                                \
    ST(1, 4, false, 1, 1, 1, 64, false)                                                                \
    ST(1, 4, false, 1, 1, 1, 32, false)                                                                \
    ST(2, 2, false, 1, 1, 1, 48, false)                                                                \
    ST(2, 2, false, 1, 1, 1, 32, false)                                                                \
    ST(2, 1, false, 1, 1, 1, 24, false)                                                                \
    ST(2, 1, false, 1, 1, 1, 16, false)                                                                \
    ST(2, 8, false, 1, 1, 1, 96, false)                                                                \
    ST(4, 4, false, 1, 1, 1, 96, false)                                                                \
    ST(2, 4, false, 0, 1, 1, 64, false)                                                                \
    ST(2, 4, false, 2, 1, 1, 0, false)                                                                 \
    ST(2, 4, false, 3, 1, 1, 0, false)                                                                 \
    ST(2, 4, false, 4, 1, 1, 0, false)                                                                 \
    ST(2, 4, false, 5, 1, 1, 0, false)                                                                 \
    ST(2, 4, false, 2, 1, 1, 64, false)                                                                \
    ST(2, 4, false, 4, 1, 1, 64, false)                                                                \
    ST(2, 4, false, 4, 1, 1, 32, false)                                                                \
    ST(2, 4, false, 4, 0, 1, 0, false)                                                                 \
    ST(2, 4, false, 4, 0, 1, 64, false)                                                                \
    ST(2, 4, false, 1, 0, 1, 64, false)                                                                \
    ST(2, 4, false, 1, 2, 1, 64, false)                                                                \
    PS(2, 4, false, 1, 0, 2, false)                                                                    \
    PS(2, 4, false, 4, 0, 2, false)                                                                    \
    PS(2, 4, false, 4, 0, 8, false)
#define SWEEP2_MIX                                                                                     \
    ST(2, 4, false, 1, 1, 1, 0, true)                                                                  \
    ST(2, 4, false, 4, 1, 1, 0, true)                                                                  \
    ST(2, 4, false, 5, 1, 1, 0, true)                                                                  \
    ST(2, 4, false, 2, 1, 1, 0, true)                                                                  \
    STL(2, 4, false, 4, 1, 1, 0, true, false)                                                          \
    STL(2, 4, false, 2, 1, 1, 0, true, false)                                                          \
    ST(1, 4, false, 4, 1, 1, 0, true)                                                                  \
    ST(4, 4, false, 4, 1, 1, 0, true)                                                                  \
    ST(8, 4, false, 4, 1, 1, 0, true)                                                                  \
    ST(2, 4, true, 4, 1, 1, 0, true)                                                                   \
    ST(4, 4, true, 4, 1, 1, 0, true)                                                                   \
    ST(2, 1, false, 4, 1, 1, 0, true)                                                                  \
    ST(2, 2, false, 4, 1, 1, 0, true)                                                                  \
    ST(2, 8, false, 4, 1, 1, 0, true)                                                                  \
    ST(2, 16, false, 4, 1, 1, 0, true)                                                                 \
    ST(2, 4, false, 4, 1, 1, 24, true)                                                                 \
    ST(2, 4, false, 4, 1, 1, 32, true)                                                                 \
    ST(2, 4, false, 4, 1, 1, 40, true)                                                                 \
    ST(4, 4, false, 4, 1, 1, 40, true)                                                                 \
    ST(4, 4, false, 4, 1, 1, 64, true)                                                                 \
    ST(8, 4, false, 4, 1, 1, 64, true)                                                                 \
    ST(2, 4, false, 4, 0, 1, 0, true)                                                                  \
    ST(2, 4, false, 4, 2, 1, 0, true)                                                                  \
    ST(2, 4, false, 4, 3, 1, 0, true)                                                                  \
    ST(2, 4, false, 4, 4, 256, 0, true)                                                                \
    PS(2, 4, false, 4, 0, 8, true)                                                                     \
    PS(2, 4, false, 4, 0, 4, true)                                                                     \
    PS(4, 4, true, 4, 2, 4, true)

// third sweep: the mix with wide loads through a wave-private LDS slab (k_mix_wide) against the best direct form, under occupancy caps
#define MW(U, WAVES, POL, LDSKB)                                                                                                       \
    if (n % (size_t(WAVES) * 128 * U) == 0 && (n / (size_t(WAVES) * 128 * U)) % 2 == 0) {                                              \
        char nm[160];                                                                                                                  \
        snprintf(nm, sizeof nm, "mix U%d x%dw wide-loads+LDS %s 2fronts lds%dK", U, WAVES, polname[POL], LDSKB);                       \
        const size_t tiles = n / (size_t(WAVES) * 128 * U);                                                                            \
        add(nm, true, [=](int i) { k_mix_wide<U, WAVES, POL, LDSKB><<<unsigned(tiles), WAVES * 64>>>(a[i % SETS], b[i % SETS], (D2*)out[i & 1], tiles); }); \
    }
#define STW(U, WAVES, POL, LDSKB, MIX)                                                                                                   \
    {                                                                                                                                  \
        char nm[160];                                                                                                                  \
        snprintf(nm, sizeof nm, "%s U%d x%dw wg-interleave %s 2fronts lds%dK wait-for-stores", MIX ? "mix" : "wr ", U, WAVES, polname[POL], LDSKB); \
        const size_t tiles = npairs / (size_t(WAVES) * 64 * U);                                                                        \
        add(nm, MIX, [=](int i) {                                                                                                      \
            k_store<U, WAVES, false, POL, 1, 1, LDSKB, MIX, true, true><<<unsigned(tiles), WAVES * 64>>>(a[i % SETS], b[i % SETS], (D2*)out[i & 1], tiles); \
        });                                                                                                                            \
    }
#define WG(POL, SCALAR, WIDE)                                                                                                          \
    {                                                                                                                                  \
        char nm[160];                                                                                                                  \
        snprintf(nm, sizeof nm, "%s U2 x4w %s %s 2fronts", SCALAR ? "sca" : "mix", WIDE ? "workgroup-wide loads + LDS" : "direct narrow loads", polname[POL]); \
        const size_t tiles = n / 1024;                                                                                                 \
        vs.push_back(Variant{nm, [=](int i) { k_wgwide<POL, SCALAR, WIDE><<<unsigned(tiles), 256>>>(a[i % SETS], b[i % SETS], (D2*)out[i & 1], tiles); }, \
                             (SCALAR ? 9.0 : 11.0) * double(n), !SCALAR, {}});                                                         \
    }
#define SWEEP3                          \
    STW(2, 4, 4, 0, true)               \
    STW(2, 4, 1, 0, true)               \
    STW(2, 4, 4, 0, false)              \
    STW(2, 4, 1, 0, false)              \
    STW(2, 4, 4, 64, false)             \
    ST(2, 4, false, 4, 1, 1, 0, false)  \
    ST(2, 4, false, 4, 1, 1, 64, false) \
    ST(2, 4, false, 4, 1, 1, 0, true)   \
    ST(2, 4, false, 1, 1, 1, 0, true)   \
    MW(2, 4, 4, 0)                      \
    MW(2, 4, 1, 0)                      \
    MW(2, 4, 4, 16)                     \
    MW(2, 4, 4, 24)                     \
    MW(2, 4, 4, 32)                     \
    MW(4, 4, 4, 0)                      \
    MW(4, 4, 4, 16)                     \
    MW(4, 4, 4, 24)                     \
    MW(4, 4, 4, 32)                     \
    MW(4, 4, 4, 48)                     \
    MW(4, 4, 1, 32)                     \
    MW(4, 2, 4, 0)                      \
    MW(4, 2, 4, 16)                     \
    MW(4, 2, 4, 24)                     \
    MW(4, 1, 4, 0)                      \
    MW(4, 1, 4, 8)                      \
    MW(4, 8, 4, 0)                      \
    MW(4, 8, 4, 48)                     \
    MW(2, 8, 4, 0)                      \
    MW(2, 2, 4, 0)

#ifndef EC_STORE_SWEEP
#define EC_STORE_SWEEP 2
#endif
#if EC_STORE_SWEEP == 1
    SWEEP1(false)
    SWEEP1(true)
#elif EC_STORE_SWEEP == 4
    WG(4, false, false) WG(4, false, true) WG(1, false, true) WG(4, true, false) WG(4, true, true) WG(1, true, false) WG(1, true, true)
    vs.push_back(Variant{"sca LIB k_binop_scalar_direct<u8, Mul, 2> (the library's kernel, its own stores), all loads nt",
                         [=](int i) { k_binop_scalar_direct<uint8_t, EC_MUL, 2, true, true><<<unsigned((n / 2 + 511) / 512), 256>>>(a[i % SETS], 2.0, out[i & 1], n, 0u); }, 9.0 * double(n), false, {}});
    vs.push_back(Variant{"sca LIB k_binop_scalar_direct<u8, Mul, 2>, operand cacheable (policy bit 0)",
                         [=](int i) { k_binop_scalar_direct<uint8_t, EC_MUL, 2, true, true><<<unsigned((n / 2 + 511) / 512), 256>>>(a[i % SETS], 2.0, out[i & 1], n, 1u << 8); }, 9.0 * double(n), false, {}});
    vs.push_back(Variant{"sca library-shaped tile, NaN rule per cell (cell_op<Mul, true>)", [=](int i) { k_scalar_lib_shape<true><<<unsigned(n / 1024), 256>>>(a[i % SETS], 2.0, out[i & 1], n); }, 9.0 * double(n), false, {}});
    vs.push_back(Variant{"sca library-shaped tile, no NaN rule (cell_op<Mul, false>)", [=](int i) { k_scalar_lib_shape<false><<<unsigned(n / 1024), 256>>>(a[i % SETS], 2.0, out[i & 1], n); }, 9.0 * double(n), false, {}});
    vs.push_back(Variant{"mix LIB k_binop_direct<u8, u16, Add, 2> (the library's kernel), all loads nt",
                         [=](int i) { k_binop_direct<uint8_t, uint16_t, EC_ADD, 2, true, true><<<unsigned((n / 2 + 511) / 512), 256>>>(a[i % SETS], b[i % SETS], out[i & 1], n, 0u); }, 11.0 * double(n), false, {}});
#elif EC_STORE_SWEEP == 5
    ST(2, 4, false, 4, 1, 1, 0, true)
    ST(2, 4, false, 1, 1, 1, 0, true)
    unsigned* tile_counter;
    CK(hipMalloc(&tile_counter, 4));
#define PP(U, D, ASSIGN, BPC)                                                                                                          \
    if (npairs % (size_t(256) * U) == 0 && npairs * 16 <= 0xffffffffull) {                                                             \
        char nm[160];                                                                                                                  \
        snprintf(nm, sizeof nm, "mix U%d x4w pipelined persistent D%d %s %d/CU %s", U, D,                                              \
                 ASSIGN == 0 ? "grid-stride" : ASSIGN == 1 ? "grid-stride-2fronts" : ASSIGN == 2 ? "chunks-atomic-queue" : "chunks-grid-stride", BPC, "nt+sc1");                                      \
        const size_t tiles = npairs / (size_t(256) * U);                                                                               \
        add(nm, true, [=](int i) {                                                                                                     \
            if (ASSIGN == 2) k_set<<<1, 1>>>(tile_counter, unsigned(cus * BPC));                                                                \
            k_mix_pipe<U, D, ASSIGN><<<unsigned(cus * BPC), 256>>>(a[i % SETS], b[i % SETS], (D2*)out[i & 1], tiles, tile_counter); \
        });                                                                                                                            \
    }
    PP(2, 2, 0, 8) PP(2, 2, 1, 8) PP(2, 2, 2, 8) PP(2, 2, 3, 8)
    PP(2, 2, 0, 4) PP(2, 2, 2, 4) PP(2, 2, 0, 2) PP(2, 2, 2, 2)
    PP(2, 3, 0, 8) PP(2, 3, 2, 8) PP(2, 3, 0, 4) PP(2, 3, 2, 4) PP(2, 3, 2, 2)
    PP(2, 4, 0, 8) PP(2, 4, 2, 8) PP(2, 4, 3, 8) PP(2, 4, 0, 4) PP(2, 4, 2, 4) PP(2, 4, 3, 4) PP(2, 4, 2, 2)
    PP(1, 2, 0, 8) PP(1, 4, 0, 8) PP(1, 4, 2, 8) PP(1, 4, 2, 4)
    PP(4, 2, 0, 8) PP(4, 2, 2, 4) PP(4, 2, 2, 2)
    PP(2, 1, 0, 8)
#elif EC_STORE_SWEEP == 6
    ST(2, 4, false, 4, 1, 1, 0, true)
    ST(2, 4, false, 1, 1, 1, 0, true)
    ST(2, 4, false, 0, 1, 1, 0, true)
    ST(2, 4, false, 2, 1, 1, 0, true)
    ST(2, 4, false, 4, 1, 1, 24, true)
    ST(4, 4, false, 4, 1, 1, 0, true)
    ST(2, 4, false, 4, 1, 1, 0, false)
    ST(2, 4, false, 1, 1, 1, 0, false)
    ST(2, 4, false, 0, 1, 1, 0, false)
    ST(2, 4, false, 4, 1, 1, 64, false)
    ST(2, 4, false, 1, 1, 1, 64, false)
    ST(2, 4, false, 0, 1, 1, 64, false)
#elif EC_STORE_SWEEP == 7
#define SW(U, WAVES, POL, LDSKB)                                                                                                       \
    if (n % (size_t(WAVES) * 128 * U) == 0 && (n / (size_t(WAVES) * 128 * U)) % 2 == 0) {                                              \
        char nm[160];                                                                                                                  \
        snprintf(nm, sizeof nm, "sca U%d x%dw wide-loads+LDS %s 2fronts lds%dK", U, WAVES, polname[POL], LDSKB);                       \
        const size_t tiles = n / (size_t(WAVES) * 128 * U);                                                                            \
        vs.push_back(Variant{nm, [=](int i) { k_sca_wide<U, WAVES, POL, LDSKB><<<unsigned(tiles), WAVES * 64>>>(a[i % SETS], (D2*)out[i & 1], tiles); }, 9.0 * double(n), false, {}}); \
    }
    WG(4, true, false)
    SW(2, 4, 4, 0) SW(2, 4, 4, 16) SW(2, 4, 4, 24)
    SW(4, 4, 4, 0) SW(4, 4, 4, 16) SW(4, 4, 4, 24) SW(4, 4, 4, 32) SW(4, 4, 4, 40) SW(4, 4, 4, 48) SW(4, 4, 4, 64)
    SW(8, 4, 4, 0) SW(8, 4, 4, 16) SW(8, 4, 4, 24) SW(8, 4, 4, 32) SW(8, 4, 4, 40) SW(8, 4, 4, 48) SW(8, 4, 4, 64) SW(8, 4, 4, 96)
    SW(8, 2, 4, 0) SW(8, 2, 4, 16) SW(8, 2, 4, 24) SW(8, 2, 4, 32) SW(8, 2, 4, 48)
    SW(4, 2, 4, 16) SW(4, 2, 4, 24) SW(4, 2, 4, 32)
    SW(8, 1, 4, 8) SW(8, 1, 4, 12) SW(8, 1, 4, 16) SW(8, 1, 4, 24)
    SW(8, 4, 1, 48) SW(8, 4, 1, 32)
    vs.push_back(Variant{"sca LIB k_binop_scalar_direct<u8, Mul, 2> without the NaN rule (what the library launches), all loads nt",
                         [=](int i) { k_binop_scalar_direct<uint8_t, EC_MUL, 2, true, true, false><<<unsigned((n / 2 + 511) / 512), 256>>>(a[i % SETS], 2.0, out[i & 1], n, 0u); }, 9.0 * double(n), false, {}});
#elif EC_STORE_SWEEP == 8
    ST(2, 4, false, 4, 1, 1, 0, true)
#define M8(WAVES, LDSKB, CLOBBER)                                                                                                      \
    if (n % (size_t(WAVES) * 1024) == 0 && (n / (size_t(WAVES) * 1024)) % 2 == 0) {                                                    \
        char nm[160];                                                                                                                  \
        snprintf(nm, sizeof nm, "mix U8 x%dw 16B/lane loads+LDS nt+sc1%s 2fronts lds%dK", WAVES, CLOBBER ? "" : " (no clobber)", LDSKB); \
        const size_t tiles = n / (size_t(WAVES) * 1024);                                                                               \
        add(nm, true, [=](int i) { k_mix_wide8<WAVES, LDSKB, CLOBBER><<<unsigned(tiles), WAVES * 64>>>(a[i % SETS], b[i % SETS], (D2*)out[i & 1], tiles); }); \
    }
    M8(4, 0, false) M8(4, 16, false) M8(4, 24, false) M8(4, 32, false) M8(4, 40, false) M8(4, 48, false) M8(4, 64, false)
    M8(2, 0, false) M8(2, 8, false) M8(2, 16, false) M8(2, 24, false) M8(2, 32, false)
    M8(1, 0, false) M8(1, 4, false) M8(1, 8, false) M8(1, 12, false) M8(1, 16, false)
    M8(4, 32, true) M8(2, 16, true)
    MW(4, 4, 4, 32) MW(4, 4, 4, 24)
    vs.push_back(Variant{"mix LIB k_binop_direct<u8, u16, Add, 2> (the library's kernel), all loads nt",
                         [=](int i) { k_binop_direct<uint8_t, uint16_t, EC_ADD, 2, true, true><<<unsigned((n / 2 + 511) / 512), 256>>>(a[i % SETS], b[i % SETS], out[i & 1], n, 0u); }, 11.0 * double(n), false, {}});
#elif EC_STORE_SWEEP == 3
    SWEEP3
#else
    SWEEP2_WRITE
    SWEEP2_MIX
#endif
    add("ref hipMemsetD32Async (the runtime's fill kernel), 8 B/cell", false, [=](int i) { CK(hipMemsetD32Async((hipDeviceptr_t)out[i & 1], 0x3ff00000, n * 2, 0)); });
    {
        const size_t tiles = n / 2 / (256 * 2);
        add("ref read 16 B/lane nt U2 two fronts, 8 B/cell", false, [=](int i) { k_read_tile<2><<<unsigned(tiles), 256>>>((const u32x4*)out[i & 1], sink); });
    }
    {   // n bytes (1 B/cell) read from 8 different places of the output buffers in rotation (all from HBM)
        const size_t g16 = n / 16;
        auto src = [=](int i) { return (const u32x4*)((const char*)out[i & 1] + size_t((i >> 1) & 3) * n); };
#define RD_TILE(BLOCK, U, FRONTS)                                                                                              \
        if (g16 % (size_t(BLOCK) * U) == 0)                                                                                    \
            vs.push_back(Variant{std::string("rd1 tile/WG " #BLOCK "thr U" #U) + (FRONTS ? " 2fronts" : " linear") + ", 1 B/cell", \
                                 [=](int i) { k_read_small_tile<BLOCK, U, FRONTS><<<unsigned(g16 / (size_t(BLOCK) * U)), BLOCK>>>(src(i), sink); }, double(n), false, {}});
#define RD_PERS(BLOCK, U, BPC)                                                                                                 \
        if (g16 % (size_t(BLOCK) * U) == 0)                                                                                    \
            vs.push_back(Variant{"rd1 persistent " #BLOCK "thr U" #U " " #BPC "/CU, 1 B/cell",                                 \
                                 [=](int i) { k_read_small_persistent<BLOCK, U><<<unsigned(cus * BPC), BLOCK>>>(src(i), g16 / (size_t(BLOCK) * U), sink); }, double(n), false, {}});
        if (!only || strstr("rd1", only) || strstr(only, "rd1")) {
            RD_TILE(256, 2, true)
            RD_TILE(256, 4, true)
            RD_TILE(256, 8, true)
            RD_TILE(512, 8, true)
            RD_TILE(512, 8, false)
            RD_TILE(512, 4, true)
            RD_PERS(512, 8, 4)
            RD_PERS(512, 8, 2)
            RD_PERS(256, 8, 8)
            RD_PERS(512, 4, 4)
            RD_PERS(1024, 8, 2)
        }
    }
    if (only && !strcmp(only, "wave-life")) {
        // not a variant: a one-off measurement, printed and done
        const size_t tiles = npairs / 512;
        unsigned* stats;
        const size_t nst = (tiles / 64 + 1) * 4 * 3;
        CK(hipMalloc(&stats, nst * 4));
        for (int pol : {1, 4}) {
            CK(hipMemset(stats, 0, nst * 4));
            for (int i = 0; i < 40; ++i) {
                if (pol == 1) k_mix_timed<1><<<unsigned(tiles), 256>>>(a[i % SETS], b[i % SETS], (D2*)out[i & 1], tiles, stats);
                else k_mix_timed<4><<<unsigned(tiles), 256>>>(a[i % SETS], b[i % SETS], (D2*)out[i & 1], tiles, stats);
            }
            CK(hipDeviceSynchronize());
            std::vector<unsigned> h(nst);
            CK(hipMemcpy(h.data(), stats, nst * 4, hipMemcpyDeviceToHost));
            double sum[3] = {0, 0, 0};
            std::vector<unsigned> col[3];
            size_t cnt = 0;
            for (size_t w = 0; w + 2 < nst; w += 3) {
                if (h[w] == 0 && h[w + 1] == 0 && h[w + 2] == 0) continue;
                for (int k = 0; k < 3; ++k) { sum[k] += h[w + k]; col[k].push_back(h[w + k]); }
                ++cnt;
            }
            printf("wave life, mix with %s stores (%zu waves sampled in the last launch; units of 10 ns):\n", polname[pol], cnt);
            const char* what[3] = {"start -> operands back", "operands back -> stores issued", "stores issued -> stores acknowledged"};
            for (int k = 0; k < 3; ++k) {
                std::sort(col[k].begin(), col[k].end());
                printf("  %-38s mean %7.1f  median %6u  p10 %6u  p90 %6u\n", what[k], sum[k] / cnt, col[k][cnt / 2], col[k][cnt / 10], col[k][cnt * 9 / 10]);
            }
        }
        return 0;
    }
    if (vs.empty()) {
        fprintf(stderr, "no variant matches '%s'\n", only ? only : "");
        return 2;
    }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<unsigned long long> sums(vs.size(), 0);
    unsigned long long rng = 12345;
    int launch_no = 0;
    // clock ramp: the first tens of milliseconds after idle run slow
    for (int i = 0; i < 120; ++i) vs[0].launch(launch_no++);
    CK(hipDeviceSynchronize());
    for (int round = -1; round < rounds; ++round) {
        std::vector<size_t> order(vs.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        if (round >= 0 && !only)
            for (size_t i = order.size(); i > 1; --i) {
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                std::swap(order[i - 1], order[(rng >> 33) % i]);
            }
        for (size_t oi = 0; oi < order.size(); ++oi) {
            Variant& v = vs[order[oi]];
            if (round < 0) {  // correctness pass: every variant must write every cell (operand set 0, output 0)
                fprintf(stderr, "check %s\n", v.name.c_str());
                CK(hipMemset(out[0], 0xEE, n * 8));
                v.launch(0);
                CK(hipGetLastError());
                CK(hipMemset(acc, 0, 8));
                k_checksum<<<2048, 256>>>((const uint64_t*)out[0], n, acc);
                CK(hipMemcpy(&sums[order[oi]], acc, 8, hipMemcpyDeviceToHost));
                CK(hipDeviceSynchronize());
                continue;
            }
            for (int i = 0; i < 2; ++i) v.launch(launch_no++);
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) v.launch(launch_no++);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms / iters);
        }
    }
    // checksums: all pure writes must agree with each other, all mixes with each other
    unsigned long long ref_wr = 0, ref_mix = 0;
    for (size_t vi = 0; vi < vs.size(); ++vi)
        if (vs[vi].name.rfind("ref", 0) != 0 && vs[vi].name.rfind("rd1", 0) != 0 && vs[vi].name.rfind("sca", 0) != 0) (vs[vi].mix ? ref_mix : ref_wr) = sums[vi];
    printf("%-74s %9s %9s %9s %9s %8s  %s\n", "variant", "med_ms", "min_ms", "max_ms", "GB/s", "of8TB/s", "cells");
    int bad = 0;
    for (size_t vi = 0; vi < vs.size(); ++vi) {
        Variant& v = vs[vi];
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        const double gbs = v.bytes / (med * 1e-3) / 1e9;
        const bool is_ref = v.name.rfind("ref", 0) == 0 || v.name.rfind("rd1", 0) == 0 || v.name.rfind("sca", 0) == 0;
        const bool ok = is_ref || sums[vi] == (v.mix ? ref_mix : ref_wr);
        bad += !ok;
        printf("%-74s %9.4f %9.4f %9.4f %9.1f %8.4f  %s\n", v.name.c_str(), med, v.ms[0], v.ms.back(), gbs, gbs / 8000.0,
               is_ref ? "-" : ok ? "ok" : "CHECKSUM DIFFERS");
    }
    return bad ? 1 : 0;
}
