// tune_reduce.hip — A/B harness for the 1 B/cell reductions (dev tool): count the 1-bytes of a 16384² mask.
//   A   the library's round-1 shape: 256-thread workgroups, grid capped at 8/CU, U = 4 guarded loads, + finalize launch
//   A2  same two launches; full tiles take a guard-free path with all U loads in flight; U and cap swept
//   A3  one tile per workgroup, straight-line (no grid-stride loop), + finalize over all partials
//   S*  ONE launch: partial stored write-through (sc1), s_waitcnt, relaxed agent-scope ticket; the workgroup whose
//       ticket is last folds the partials with sc1 loads that are ALL in flight before the first add
//       (round 1's variant C folded with a dependent chain of 8 atomic loads per lane: +16 µs)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_reduce.hip -o tools/tune_reduce && ./tools/tune_reduce
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t wave_sum(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t pop16(u32x4 x) {
    return __builtin_popcount(x.x & 0x01010101u) + __builtin_popcount(x.y & 0x01010101u) +
           __builtin_popcount(x.z & 0x01010101u) + __builtin_popcount(x.w & 0x01010101u);
}

template <int BLOCK>
__device__ __forceinline__ uint64_t block_sum(uint64_t c) {
    c = wave_sum(c);
    __shared__ uint64_t s[BLOCK / 64];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    uint64_t t = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) t += s[w];
    return t;
}

// grid-stride over tiles of BLOCK*U groups of 16 bytes
template <int BLOCK, int U, bool FAST>
__device__ __forceinline__ uint64_t count_stride(const uint8_t* __restrict__ m, size_t n) {
    const size_t ngroups = n / 16;
    constexpr size_t TILE = size_t(BLOCK) * U;
    const size_t ntiles = (ngroups + TILE - 1) / TILE;
    const u32x4* __restrict__ mv = reinterpret_cast<const u32x4*>(m);
    uint32_t c = 0;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t base = tile * TILE + threadIdx.x;
        if (FAST && tile * TILE + TILE <= ngroups) {
            u32x4 x[U];
#pragma unroll
            for (int j = 0; j < U; ++j) x[j] = __builtin_nontemporal_load(mv + base + size_t(j) * BLOCK);
#pragma unroll
            for (int j = 0; j < U; ++j) c += pop16(x[j]);
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const size_t g = base + size_t(j) * BLOCK;
                if (g < ngroups) c += pop16(__builtin_nontemporal_load(mv + g));
            }
        }
    }
    return c;
}

template <int BLOCK, int U, bool FAST>
__global__ __launch_bounds__(BLOCK) void k_partials(const uint8_t* m, size_t n, uint64_t* partials) {
    const uint64_t c = block_sum<BLOCK>(count_stride<BLOCK, U, FAST>(m, n));
    if (threadIdx.x == 0) partials[blockIdx.x] = c;
}

// one tile per workgroup, two fronts
template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void k_partials_tile(const uint8_t* m, size_t n, uint64_t* partials) {
    const size_t ngroups = n / 16;
    constexpr size_t TILE = size_t(BLOCK) * U;
    const size_t b = blockIdx.x, nb = gridDim.x;
    const size_t tile = (b & 1) ? nb - 1 - (b >> 1) : (b >> 1);
    const size_t base = tile * TILE + threadIdx.x;
    const u32x4* __restrict__ mv = reinterpret_cast<const u32x4*>(m);
    uint32_t c = 0;
    if (tile * TILE + TILE <= ngroups) {
        u32x4 x[U];
#pragma unroll
        for (int j = 0; j < U; ++j) x[j] = __builtin_nontemporal_load(mv + base + size_t(j) * BLOCK);
#pragma unroll
        for (int j = 0; j < U; ++j) c += pop16(x[j]);
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t g = base + size_t(j) * BLOCK;
            if (g < ngroups) c += pop16(__builtin_nontemporal_load(mv + g));
        }
    }
    const uint64_t t = block_sum<BLOCK>(c);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

template <int PER>
__global__ __launch_bounds__(1024) void k_finalize(const uint64_t* __restrict__ partials, int nparts, uint64_t* out) {
    uint64_t v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = threadIdx.x + j * 1024;
        v[j] = i < nparts ? partials[i] : 0ull;
    }
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) c += v[j];
    c = block_sum<1024>(c);
    if (threadIdx.x == 0) out[0] = c;
}

// finalize with BLOCK threads, every thread's PER loads in flight before the first add
template <int BLOCK, int PER>
__global__ __launch_bounds__(BLOCK) void k_finalize_b(const uint64_t* __restrict__ partials, int nparts, uint64_t* out) {
    uint64_t v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = threadIdx.x + j * BLOCK;
        v[j] = i < nparts ? partials[i] : 0ull;
    }
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) c += v[j];
    c = block_sum<BLOCK>(c);
    if (threadIdx.x == 0) out[0] = c;
}

// single launch
template <int BLOCK, int U, int FOLD_PER>
__global__ __launch_bounds__(BLOCK) void k_single(const uint8_t* m, size_t n, uint64_t* partials, unsigned* ticket, uint64_t* out) {
    const uint64_t c = block_sum<BLOCK>(count_stride<BLOCK, U, true>(m, n));
    __shared__ unsigned last;
    if (threadIdx.x == 0) {
        __hip_atomic_store(partials + blockIdx.x, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store sc1: write-through
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0);  // the store has left before the ticket is drawn
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {  // workgroup-uniform
        uint64_t v[FOLD_PER];
#pragma unroll
        for (int j = 0; j < FOLD_PER; ++j) {  // all loads in flight before the first add
            const int i = threadIdx.x + j * BLOCK;
            v[j] = i < int(gridDim.x) ? __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        uint64_t t = 0;
#pragma unroll
        for (int j = 0; j < FOLD_PER; ++j) t += v[j];
        __syncthreads();  // block_sum's shared array is reused
        t = block_sum<BLOCK>(t);
        if (threadIdx.x == 0) {
            out[0] = t;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        }
    }
}

__global__ void k_fill(uint8_t* m, size_t n, uint64_t seed) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t x = (i + seed) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29;
        m[i] = (x % 100) >= 30;
    }
}

struct Variant {
    const char* name;
    void (*run)(const uint8_t*, size_t, uint64_t*, unsigned*, uint64_t*);
};

static size_t tiles_of(size_t n, size_t tile_groups) { return (n / 16 + tile_groups - 1) / tile_groups; }

#define TWO(NAME, BLOCK, U, FAST, GRID)                                                                         \
    {NAME, [](const uint8_t* m, size_t n, uint64_t* p, unsigned*, uint64_t* o) {                                \
         const int g = (int)std::min<size_t>(GRID, tiles_of(n, size_t(BLOCK) * U));                            \
         k_partials<BLOCK, U, FAST><<<g, BLOCK>>>(m, n, p);                                                     \
         k_finalize<4><<<1, 1024>>>(p, g, o);                                                                   \
     }}
#define TWOF(NAME, FB, FPER)                                                                                    \
    {NAME, [](const uint8_t* m, size_t n, uint64_t* p, unsigned*, uint64_t* o) {                                \
         const int g = (int)std::min<size_t>(1024, tiles_of(n, size_t(512) * 8));                              \
         k_partials<512, 8, true><<<g, 512>>>(m, n, p);                                                         \
         k_finalize_b<FB, FPER><<<1, FB>>>(p, g, o);                                                            \
     }}
#define TILE(NAME, BLOCK, U)                                                                                    \
    {NAME, [](const uint8_t* m, size_t n, uint64_t* p, unsigned*, uint64_t* o) {                                \
         const int g = (int)tiles_of(n, size_t(BLOCK) * U);                                                     \
         k_partials_tile<BLOCK, U><<<g, BLOCK>>>(m, n, p);                                                      \
         k_finalize<32><<<1, 1024>>>(p, g, o);                                                                  \
     }}
#define ONE(NAME, BLOCK, U, GRID)                                                                               \
    {NAME, [](const uint8_t* m, size_t n, uint64_t* p, unsigned* t, uint64_t* o) {                              \
         const int g = (int)std::min<size_t>(GRID, tiles_of(n, size_t(BLOCK) * U));                            \
         k_single<BLOCK, U, (GRID + BLOCK - 1) / BLOCK><<<g, BLOCK>>>(m, n, p, t, o);                           \
     }}

int main(int argc, char** argv) {
    const size_t side = argc > 1 ? atol(argv[1]) : 16384;
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    const size_t n = side * side;
    uint8_t* m;
    uint64_t *partials, *out;
    unsigned* ticket;
    CK(hipMalloc(&m, n));
    CK(hipMalloc(&partials, 32768 * 8));
    CK(hipMalloc(&out, 8));
    CK(hipMalloc(&ticket, 4));
    CK(hipMemset(ticket, 0, 4));
    k_fill<<<2048, 256>>>(m, n, 977);
    const Variant vs[] = {
        TWO("A   2 launches 256thr U4 guarded cap2048 (round 1)", 256, 4, false, 2048),
        TWO("A2  2 launches 256thr U4 fast cap2048", 256, 4, true, 2048),
        TWO("A2  2 launches 256thr U8 fast cap2048", 256, 8, true, 2048),
        TWO("A2  2 launches 256thr U8 fast cap1024", 256, 8, true, 1024),
        TWO("A2  2 launches 256thr U8 fast cap4096", 256, 8, true, 4096),
        TWO("A2  2 launches 256thr U16 fast cap2048", 256, 16, true, 2048),
        TWO("A2  2 launches 512thr U8 fast cap1024", 512, 8, true, 1024),
        TWO("A2  2 launches 1024thr U8 fast cap256", 1024, 8, true, 256),
        TWO("A2  2 launches 1024thr U8 fast cap512", 1024, 8, true, 512),
        TWOF("F   512thr U8 cap1024 + finalize 64 thr x16", 64, 16),
        TWOF("F   512thr U8 cap1024 + finalize 128 thr x8", 128, 8),
        TWOF("F   512thr U8 cap1024 + finalize 256 thr x4", 256, 4),
        TWOF("F   512thr U8 cap1024 + finalize 512 thr x2", 512, 2),
        TILE("A3  2 launches tile/WG 256thr U8", 256, 8),
        TILE("A3  2 launches tile/WG 256thr U16", 256, 16),
        TILE("A3  2 launches tile/WG 512thr U16", 512, 16),
        ONE("S   1 launch 256thr U8 grid2048", 256, 8, 2048),
        ONE("S   1 launch 256thr U8 grid1024", 256, 8, 1024),
        ONE("S   1 launch 512thr U8 grid1024", 512, 8, 1024),
        ONE("S   1 launch 512thr U8 grid512", 512, 8, 512),
        ONE("S   1 launch 1024thr U8 grid256", 1024, 8, 256),
        ONE("S   1 launch 1024thr U8 grid512", 1024, 8, 512),
        ONE("S   1 launch 1024thr U4 grid512", 1024, 4, 512),
    };
    const int nv = sizeof vs / sizeof vs[0];
    uint64_t expect = 0;
    vs[0].run(m, n, partials, ticket, out);
    CK(hipMemcpy(&expect, out, 8, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> best(nv, 1e9f), sum(nv, 0.f);
    std::vector<int> wrong(nv, 0);
    const int iters = 400;
    for (int r = 0; r < rounds; ++r) {
        for (int k = 0; k < nv; ++k) {
            const int v = (k * 7 + r * 3) % nv;  // a different order every round (nv = 23 is prime to 7)
            for (int i = 0; i < 200; ++i) vs[v].run(m, n, partials, ticket, out);  // clocks up, caches in steady state
            CK(hipMemsetAsync(out, 0, 8));
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) vs[v].run(m, n, partials, ticket, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters;
            best[v] = std::min(best[v], ms);
            sum[v] += ms;
            uint64_t got = 0;
            CK(hipMemcpy(&got, out, 8, hipMemcpyDeviceToHost));
            if (got != expect) ++wrong[v];
        }
    }
    printf("count of 1-bytes, %zux%zu mask (%zu B), %d rounds x %d launches each, expected %llu\n", side, side, n, rounds, iters,
           (unsigned long long)expect);
    for (int v = 0; v < nv; ++v)
        printf("%-56s mean %.4f ms  best %.4f ms  %.0f GB/s (%.3f of 8 TB/s)  wrong %d\n", vs[v].name, sum[v] / rounds, best[v],
               n / (sum[v] / rounds * 1e-3) / 1e9, n / (sum[v] / rounds * 1e-3) / 1e9 / 8000.0, wrong[v]);
    return 0;
}
