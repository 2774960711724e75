// tune_binop.hip — A/B harness for the u8 ÷ u16 -> f64 kernel shape (dev tool, not part of the library).
//
// One experimental kernel (k_exp) with every knob that was swept in round 1 — tile depth U, workgroup
// size, store cache policy, non-temporal loads, occupancy cap (LDS reservation), store pacing
// (s_sleep), workgroup->tile order (single front / XCD-contiguous / two fronts), an unscaled
// integer-operand divide — next to the library's own kernel on the same buffers and to pure
// read / write / copy reference kernels.  Variants run in randomized order over interleaved rounds
// (guide §5.4 rule 24); medians, minima and maxima are printed.  The logs of the five sweeps are
// profiles/r01/tune_binop_v1..v5.log; edit the EXPX(...) list below to re-run a sweep.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ierased-cells_amd/csrc tools/tune_binop.hip -o tools/tune_binop
//   ./tools/tune_binop [side=16384] [rounds=11] [iters=10]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
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

// experimental: divide for operands that cannot over/underflow (integers): no v_div_scale / v_div_fmas
__device__ __forceinline__ double int_div(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    double q = a * r;
    double rem = __builtin_fma(-b, q, a);
    q = __builtin_fma(rem, r, q);
    q = __builtin_amdgcn_div_fixup(q, b, a);
    return (q != q) ? bits_f64(kNegQNaN) : q;
}


// store policies: 0 plain, 1 nt (builtin), 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 sc0, 6 nt sc0 sc1
template <int POL>
__device__ __forceinline__ void store16(D2* p, D2 v) {
    if constexpr (POL == 0) *p = v;
    else if constexpr (POL == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int U, int BLOCK, bool WAVE_CONTIG, int STPOL, bool NTL, int OP, int LDSKB = 0, int SLEEP = 0, int PERM = 0>
__global__ __launch_bounds__(BLOCK) void k_exp(const uint8_t* __restrict__ l, const uint16_t* __restrict__ r,
                                               double* __restrict__ out, size_t n) {
    using L2 = vec<uint8_t, 2>;
    using R2 = vec<uint16_t, 2>;
    if constexpr (LDSKB > 0) {  // occupancy cap: reserve LDS so fewer workgroups fit per CU
        __shared__ volatile uint32_t pad[LDSKB * 256];
        if (threadIdx.x == 0) pad[0] = 1;
    }
    const size_t npairs = n >> 1;
    constexpr size_t TILE = size_t(BLOCK) * U;
    size_t tile = blockIdx.x;  // one block per tile; host guarantees npairs % TILE == 0
    if constexpr (PERM == 1) {  // XCD-contiguous: blocks b, b+8, ... (same XCD) walk one contiguous eighth of the raster
        const size_t per = gridDim.x / 8;
        tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    } else if constexpr (PERM == 2) {  // two fronts: even blocks from the start, odd blocks from the end
        tile = (blockIdx.x & 1) ? gridDim.x - 1 - (blockIdx.x >> 1) : (blockIdx.x >> 1);
    } else if constexpr (PERM == 3) {  // two fronts, decoupled from XCD parity: groups of 8 consecutive blocks alternate
        const size_t g = blockIdx.x >> 3, r = blockIdx.x & 7, k = (g >> 1) * 8 + r;
        tile = (g & 1) ? gridDim.x - 1 - k : k;
    } else if constexpr (PERM == 4) {  // four fronts (b & 3): quarters walked forward
        const size_t per = gridDim.x / 4;
        tile = (blockIdx.x & 3) * per + (blockIdx.x >> 2);
    } else if constexpr (PERM == 5) {  // four fronts: two from the start of each half, two from the ends
        const size_t half = gridDim.x / 2, f = blockIdx.x & 3, k = blockIdx.x >> 2;
        tile = f == 0 ? k : f == 1 ? half - 1 - k : f == 2 ? half + k : gridDim.x - 1 - k;
    }
    size_t base, stride;
    if constexpr (WAVE_CONTIG) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        base = tile * TILE + size_t(wave) * (64 * U) + lane;
        stride = 64;
    } else {
        base = tile * TILE + threadIdx.x;
        stride = BLOCK;
    }
    if (base + (U - 1) * stride >= npairs) return;
    const L2* lp = reinterpret_cast<const L2*>(l);
    const R2* rp = reinterpret_cast<const R2*>(r);
    D2* op = reinterpret_cast<D2*>(out);
    L2 a[U];
    R2 b[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        a[j] = load_vec<NTL>(lp + base + j * stride);
        b[j] = load_vec<NTL>(rp + base + j * stride);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
        D2 o;
        if constexpr (OP == 7) {
            o.x = int_div(to_f64(a[j].x), to_f64(b[j].x));
            o.y = int_div(to_f64(a[j].y), to_f64(b[j].y));
        } else {
            o.x = cell_op<OP, false>(to_f64(a[j].x), to_f64(b[j].x));
            o.y = cell_op<OP, false>(to_f64(a[j].y), to_f64(b[j].y));
        }
        store16<STPOL>(op + base + j * stride, o);
        if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
}

// LDS-staged counterpart of the best DIRECT shape (U = 2, 256 cells per wave): the wave loads its u8 tile
// with one dword per lane and its u16 tile with one dwordx2 per lane (half the global load instructions of
// DIRECT), stages both in a wave-private LDS slab and reads back the two cells of each output slot.
template <int STPOL, bool NTL, int OP, int PERM>
__global__ __launch_bounds__(256) void k_lds2(const uint8_t* __restrict__ l, const uint16_t* __restrict__ r,
                                              double* __restrict__ out, size_t n) {
    __shared__ __attribute__((aligned(16))) unsigned char slab_a[4][256];
    __shared__ __attribute__((aligned(16))) unsigned char slab_b[4][512];
    size_t blk = blockIdx.x;
    if constexpr (PERM == 2) blk = (blockIdx.x & 1) ? gridDim.x - 1 - (blockIdx.x >> 1) : (blockIdx.x >> 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t cell0 = (blk * 4 + wave) * 256;  // host guarantees n % 1024 == 0
    if (cell0 + 256 > n) return;
    const uint32_t av = load_vec<NTL>(reinterpret_cast<const uint32_t*>(l + cell0) + lane);
    const vec<uint32_t, 2> bv = load_vec<NTL>(reinterpret_cast<const vec<uint32_t, 2>*>(r + cell0) + lane);
    reinterpret_cast<uint32_t*>(slab_a[wave])[lane] = av;
    reinterpret_cast<vec<uint32_t, 2>*>(slab_b[wave])[lane] = bv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    using L2 = vec<uint8_t, 2>;
    using R2 = vec<uint16_t, 2>;
    D2* op = reinterpret_cast<D2*>(out + cell0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const L2 a = reinterpret_cast<const L2*>(slab_a[wave])[j * 64 + lane];
        const R2 b = reinterpret_cast<const R2*>(slab_b[wave])[j * 64 + lane];
        D2 o;
        o.x = cell_op<OP, false>(to_f64(a.x), to_f64(b.x));
        o.y = cell_op<OP, false>(to_f64(a.y), to_f64(b.y));
        store16<STPOL>(op + j * 64 + lane, o);
    }
}

// "lds3": as lds2, but the tiles go from global memory straight into the wave-private LDS slab with gfx950's
// `global_load_lds_dword` (no VGPR round trip, no ds_write): LDS address = M0 base + inst offset + lane*4.
template <int STPOL, int AUX, int OP, int PERM>
__global__ __launch_bounds__(256) void k_lds3(const uint8_t* __restrict__ l, const uint16_t* __restrict__ r,
                                              double* __restrict__ out, size_t n) {
    __shared__ __attribute__((aligned(16))) unsigned char slab_a[4][256];
    __shared__ __attribute__((aligned(16))) unsigned char slab_b[4][512];
    size_t blk = blockIdx.x;
    if constexpr (PERM == 2) blk = (blockIdx.x & 1) ? gridDim.x - 1 - (blockIdx.x >> 1) : (blockIdx.x >> 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t cell0 = (blk * 4 + wave) * 256;  // host guarantees n % 1024 == 0
    if (cell0 + 256 > n) return;
    typedef const void __attribute__((address_space(1)))* gptr;
    typedef void __attribute__((address_space(3)))* lptr;
    __builtin_amdgcn_global_load_lds((gptr)(l + cell0 + lane * 4), (lptr)slab_a[wave], 4, 0, AUX);
    __builtin_amdgcn_global_load_lds((gptr)(reinterpret_cast<const unsigned char*>(r + cell0) + lane * 4), (lptr)slab_b[wave], 4, 0, AUX);
    __builtin_amdgcn_global_load_lds((gptr)(reinterpret_cast<const unsigned char*>(r + cell0) + lane * 4), (lptr)slab_b[wave], 4, 256, AUX);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    using L2 = vec<uint8_t, 2>;
    using R2 = vec<uint16_t, 2>;
    D2* op = reinterpret_cast<D2*>(out + cell0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const L2 a = reinterpret_cast<const volatile L2*>(slab_a[wave])[j * 64 + lane];
        const R2 b = reinterpret_cast<const volatile R2*>(slab_b[wave])[j * 64 + lane];
        D2 o;
        o.x = cell_op<OP, false>(to_f64(a.x), to_f64(b.x));
        o.y = cell_op<OP, false>(to_f64(a.y), to_f64(b.y));
        store16<STPOL>(op + j * 64 + lane, o);
    }
}

__global__ void k_fill(uint8_t* a, uint16_t* b, size_t n) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        a[i] = uint8_t(splitmix64(0x5EED0001ull ^ i) % 256);
        b[i] = uint16_t(1 + splitmix64(0x5EED0002ull ^ i) % 65535);
    }
}

__global__ void k_checksum(const uint64_t* p, size_t n, unsigned long long* acc) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    unsigned long long s = 0;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) s += p[i] * (i | 1);
    atomicAdd(acc, s);
}

// pure-memory references, one block per 256*U*16 B tile
template <int U>
__global__ __launch_bounds__(256) void k_copy_tile(const u32x4* __restrict__ s, u32x4* __restrict__ d) {
    size_t base = size_t(blockIdx.x) * 256 * U + threadIdx.x;
    u32x4 x[U];
#pragma unroll
    for (int j = 0; j < U; ++j) x[j] = s[base + j * 256];
#pragma unroll
    for (int j = 0; j < U; ++j) d[base + j * 256] = x[j];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_write_tile(u32x4* __restrict__ d) {
    size_t base = size_t(blockIdx.x) * 256 * U + threadIdx.x;
    u32x4 v = {1, 2, 3, 4};
#pragma unroll
    for (int j = 0; j < U; ++j) store_vec<NT>(d + base + j * 256, v);
}
template <int U>
__global__ __launch_bounds__(256) void k_read_tile(const u32x4* __restrict__ s, uint32_t* sink) {
    size_t base = size_t(blockIdx.x) * 256 * U + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < U; ++j) acc ^= s[base + j * 256];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) *sink = 1;
}

struct Variant {
    std::string name;
    std::function<void()> launch;
    double bytes;
    std::vector<float> ms;
};

int main(int argc, char** argv) {
    size_t side = argc > 1 ? strtoull(argv[1], 0, 10) : 16384;
    int rounds = argc > 2 ? atoi(argv[2]) : 5;
    int iters = argc > 3 ? atoi(argv[3]) : 10;
    const size_t n = side * side;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  n %zu cells\n", prop.gcnArchName, prop.multiProcessorCount, n);
    uint8_t* a;
    uint16_t* b;
    double* out;
    unsigned long long* acc;
    uint32_t* sink;
    CK(hipMalloc(&a, n));
    CK(hipMalloc(&b, n * 2));
    CK(hipMalloc(&out, n * 8));
    CK(hipMalloc(&acc, 8));
    CK(hipMalloc(&sink, 4));
    k_fill<<<2048, 256>>>(a, b, n);
    CK(hipDeviceSynchronize());

    std::vector<Variant> vs;
    const double b11 = 11.0 * double(n);
    auto add = [&](std::string name, double bytes, std::function<void()> f) { vs.push_back(Variant{name, f, bytes, {}}); };

#define EXP(U, BLOCK, WC, STPOL, NTL, OPV, OPN)                                                               \
    if ((n / 2) % (size_t(BLOCK) * U) == 0)                                                                  \
        add(std::string("exp ") + OPN + " U" #U " blk" #BLOCK " wc" #WC " st" #STPOL " ntl" #NTL, b11, [=]() {   \
            k_exp<U, BLOCK, WC, STPOL, NTL, OPV><<<unsigned((n / 2) / (size_t(BLOCK) * U)), BLOCK>>>(a, b, out, n); \
        })

    add("ref copy_tile U8 (4B rd + 4B wr per cell)", 8.0 * n, [=]() { k_copy_tile<8><<<unsigned(n / 4 / (256 * 8)), 256>>>((const u32x4*)out, (u32x4*)out + n / 4); });
    add("ref write_tile U8 plain (8B/cell)", 8.0 * n, [=]() { k_write_tile<8, false><<<unsigned(n / 2 / (256 * 8)), 256>>>((u32x4*)out); });
    add("ref write_tile U8 nt (8B/cell)", 8.0 * n, [=]() { k_write_tile<8, true><<<unsigned(n / 2 / (256 * 8)), 256>>>((u32x4*)out); });
    add("ref read_tile U8 (8B/cell)", 8.0 * n, [=]() { k_read_tile<8><<<unsigned(n / 2 / (256 * 8)), 256>>>((const u32x4*)out, sink); });

#define EXPX(U, BLOCK, STPOL, NTL, OPV, OPN, LDSKB, SLEEP, PERM)                                             \
    if ((n / 2) % (size_t(BLOCK) * U) == 0 && ((n / 2) / (size_t(BLOCK) * U)) % 8 == 0)                      \
        add(std::string("exp ") + OPN + " U" #U " blk" #BLOCK " ntl" #NTL " lds" #LDSKB "KB sleep" #SLEEP " perm" #PERM, b11, [=]() { \
            k_exp<U, BLOCK, false, STPOL, NTL, OPV, LDSKB, SLEEP, PERM><<<unsigned((n / 2) / (size_t(BLOCK) * U)), BLOCK>>>(a, b, out, n); \
        })
    EXPX(2, 256, 1, true, EC_DIV, "div", 0, 0, 0);
    EXPX(2, 256, 1, true, EC_DIV, "div", 0, 0, 2);
    EXPX(2, 256, 1, true, EC_DIV, "div", 0, 0, 3);
    EXPX(2, 256, 1, true, EC_DIV, "div", 0, 0, 4);
    EXPX(2, 256, 1, true, EC_DIV, "div", 0, 0, 5);
    EXPX(2, 256, 1, true, EC_ADD, "add", 0, 0, 0);
    EXPX(2, 256, 1, true, EC_ADD, "add", 0, 0, 2);
    EXPX(2, 256, 1, true, EC_ADD, "add", 0, 0, 3);
    EXPX(2, 256, 1, true, EC_ADD, "add", 0, 0, 4);
    EXPX(2, 256, 1, true, EC_ADD, "add", 0, 0, 5);
    if (n % 1024 == 0) {
        add("lds2 div U2 wave-private slab, dword/dwordx2 loads, perm2", b11, [=]() { k_lds2<1, true, EC_DIV, 2><<<unsigned(n / 1024), 256>>>(a, b, out, n); });
        add("lds2 div U2 wave-private slab, dword/dwordx2 loads, perm0", b11, [=]() { k_lds2<1, true, EC_DIV, 0><<<unsigned(n / 1024), 256>>>(a, b, out, n); });
        add("lds3 div direct-to-LDS (global_load_lds_dword), perm2", b11, [=]() { k_lds3<1, 0, EC_DIV, 2><<<unsigned(n / 1024), 256>>>(a, b, out, n); });
        add("lds3 div direct-to-LDS, nt aux=2, perm2", b11, [=]() { k_lds3<1, 2, EC_DIV, 2><<<unsigned(n / 1024), 256>>>(a, b, out, n); });
        add("lds3 add direct-to-LDS (global_load_lds_dword), perm2", b11, [=]() { k_lds3<1, 0, EC_ADD, 2><<<unsigned(n / 1024), 256>>>(a, b, out, n); });
        add("lds2 add U2 wave-private slab, dword/dwordx2 loads, perm2", b11, [=]() { k_lds2<1, true, EC_ADD, 2><<<unsigned(n / 1024), 256>>>(a, b, out, n); });
    }
    add("LIB k_binop_direct div U2 nt/nt (library kernel, same buffers)", b11, [=]() {
        k_binop_direct<uint8_t, uint16_t, EC_DIV, 2, true, true><<<unsigned((n / 2 + 511) / 512), 256>>>(a, b, out, n); });
    add("LIB k_binop_direct add U2 nt/nt (library kernel, same buffers)", b11, [=]() {
        k_binop_direct<uint8_t, uint16_t, EC_ADD, 2, true, true><<<unsigned((n / 2 + 511) / 512), 256>>>(a, b, out, n); });
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<unsigned long long> sums(vs.size(), 0);
    unsigned long long rng = 12345;
    for (int round = -1; round < rounds; ++round) {
        std::vector<size_t> order(vs.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        if (round >= 0) { for (size_t i = order.size(); i > 1; --i) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; std::swap(order[i - 1], order[(rng >> 33) % i]); } }
        for (size_t oi = 0; oi < order.size(); ++oi) {
            size_t vi = order[oi];
            Variant& v = vs[vi];
            if (round < 0) {
                CK(hipMemset(out, 0xEE, 64));
                v.launch();
                CK(hipGetLastError());
                CK(hipMemset(acc, 0, 8));
                k_checksum<<<2048, 256>>>((const uint64_t*)out, n, acc);
                CK(hipMemcpy(&sums[vi], acc, 8, hipMemcpyDeviceToHost));
                continue;
            }
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) v.launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms / iters);
        }
    }
    printf("%-52s %9s %9s %9s %9s %8s  %s\n", "variant", "med_ms", "min_ms", "max_ms", "GB/s", "%of8TB", "checksum");
    for (size_t vi = 0; vi < vs.size(); ++vi) {
        Variant& v = vs[vi];
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        double gbs = v.bytes / (med * 1e-3) / 1e9;
        printf("%-52s %9.4f %9.4f %9.4f %9.1f %7.1f%%  %016llx\n", v.name.c_str(), med, mn, v.ms.back(), gbs, gbs / 80.0, sums[vi]);
    }
    return 0;
}
