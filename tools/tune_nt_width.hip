// tune_nt_width.hip — which operand loads should be non-temporal?  (dev tool, round 3)
//
// tools/tune_nt_u8.hip showed that putting `nt` on the 2-byte-per-lane loads of a 1-byte operand stream LOSES 8-16 %
// (u8/u16 divide 0.868 -> 0.797, u8 x scalar 0.938 -> 0.786): the plain `global_load_ushort` rounds 1-2 shipped by
// accident was the better instruction.  This tool asks the same question for every load width of k_binop_direct's
// access pattern (2 cells per lane per chunk, lane-contiguous): ushort (1-byte cells), dword (2-byte), dwordx2
// (4-byte), dwordx4 (8-byte), each stream nt or plain independently; stores stay nt.  Randomised interleaved rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -Iinclude -Ierased-cells_amd/csrc \
//         tools/tune_nt_width.hip -o tools/tune_nt_width && ./tools/tune_nt_width [side] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "ec_binop_kernels.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace ecd;

// a lane's pair of cells as unsigned words of the pair's size (so that `nt` is honoured at every width)
template <int BYTES> struct pair_words;
template <> struct pair_words<2> { using type = uint16_t; };
template <> struct pair_words<4> { using type = uint32_t; };
template <> struct pair_words<8> { using type = vec<uint32_t, 2>; };
template <> struct pair_words<16> { using type = vec<uint32_t, 4>; };

template <typename T, bool NT>
struct PairLd {
    using W = typename pair_words<2 * sizeof(T)>::type;
    static __device__ __forceinline__ W load(const T* first_cell) {
        const W* p = reinterpret_cast<const W*>(first_cell);
        if constexpr (NT) return nt_load(p);
        else return plain_load(p);
    }
    static __device__ __forceinline__ double cell(W w, int k) {
        if constexpr (sizeof(T) == 1) return to_f64(static_cast<T>((uint32_t(w) >> (8 * k)) & 0xffu));
        else if constexpr (sizeof(T) == 2) return to_f64(static_cast<T>((w >> (16 * k)) & 0xffffu));
        else if constexpr (sizeof(T) == 4) { const uint32_t x = k ? w.y : w.x; return to_f64(__builtin_bit_cast(T, x)); }
        else { const uint64_t q = k ? (uint64_t(w.z) | (uint64_t(w.w) << 32)) : (uint64_t(w.x) | (uint64_t(w.y) << 32)); return to_f64(__builtin_bit_cast(T, q)); }
    }
};

template <typename L, typename R, int OP, bool NTL, bool NTR>
__global__ __launch_bounds__(kBlock) void k2(const L* __restrict__ l, const R* __restrict__ r, double* __restrict__ out, size_t n) {
    using D2 = vec<double, 2>;
    constexpr int U = 2;
    constexpr bool FP = is_fp<L>::value || is_fp<R>::value;
    constexpr bool SM = is_small_int<L>::value && is_small_int<R>::value;
    constexpr size_t TILE = size_t(kBlock) * U;
    const size_t tile = two_front_tile();
    if (tile * TILE + TILE > (n >> 1)) return;
    const size_t base = tile * TILE + threadIdx.x;
    typename PairLd<L, NTL>::W a[U];
    typename PairLd<R, NTR>::W b[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        a[j] = PairLd<L, NTL>::load(l + 2 * (base + size_t(j) * kBlock));
        b[j] = PairLd<R, NTR>::load(r + 2 * (base + size_t(j) * kBlock));
    }
    D2* op = reinterpret_cast<D2*>(out);
#pragma unroll
    for (int j = 0; j < U; ++j) {
        D2 o;
        o.x = cell_op<OP, FP, SM>(PairLd<L, NTL>::cell(a[j], 0), PairLd<R, NTR>::cell(b[j], 0));
        o.y = cell_op<OP, FP, SM>(PairLd<L, NTL>::cell(a[j], 1), PairLd<R, NTR>::cell(b[j], 1));
        nt_store(o, op + base + size_t(j) * kBlock);
    }
}

template <typename L, int OP, bool NTL>
__global__ __launch_bounds__(kBlock) void k1(const L* __restrict__ l, double s, double* __restrict__ out, size_t n) {
    using D2 = vec<double, 2>;
    constexpr int U = 2;
    constexpr size_t TILE = size_t(kBlock) * U;
    const size_t tile = two_front_tile();
    if (tile * TILE + TILE > (n >> 1)) return;
    const size_t base = tile * TILE + threadIdx.x;
    typename PairLd<L, NTL>::W a[U];
#pragma unroll
    for (int j = 0; j < U; ++j) a[j] = PairLd<L, NTL>::load(l + 2 * (base + size_t(j) * kBlock));
    D2* op = reinterpret_cast<D2*>(out);
#pragma unroll
    for (int j = 0; j < U; ++j) {
        D2 o;
        o.x = cell_op<OP, true>(PairLd<L, NTL>::cell(a[j], 0), s);
        o.y = cell_op<OP, true>(PairLd<L, NTL>::cell(a[j], 1), s);
        nt_store(o, op + base + size_t(j) * kBlock);
    }
}

// read-only stream at 16 B per lane, 8 loads in flight (the reductions' shape), nt or plain
template <bool NT>
__global__ __launch_bounds__(512) void k_read(const u32x4* __restrict__ p, size_t ngroups, uint32_t* __restrict__ sink) {
    constexpr int U = 8;
    constexpr size_t TILE = size_t(512) * U;
    uint32_t acc = 0;
    const size_t ntiles = ngroups / TILE;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t base = tile * TILE + threadIdx.x;
        u32x4 x[U];
#pragma unroll
        for (int j = 0; j < U; ++j) x[j] = NT ? nt_load(p + base + size_t(j) * 512) : plain_load(p + base + size_t(j) * 512);
#pragma unroll
        for (int j = 0; j < U; ++j) acc ^= x[j].x ^ x[j].y ^ x[j].z ^ x[j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void k_fill(uint32_t* p, size_t nwords) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nwords; i += stride)
        p[i] = 0x3f800000u | (uint32_t(splitmix64(i)) & 0x007fffffu) | 0x01010101u;  // finite floats, no zero cell of any width
}

struct Variant {
    std::string name;
    std::function<void()> launch;
    double bpc;
    std::vector<float> ms;
};

int main(int argc, char** argv) {
    const size_t side = argc > 1 ? strtoull(argv[1], nullptr, 10) : 16384;
    const int rounds = argc > 2 ? atoi(argv[2]) : 9;
    const size_t n = side * side;
    void *a, *b;
    double* out;
    uint32_t* sink;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&sink, 64));
    k_fill<<<4096, 256>>>(static_cast<uint32_t*>(a), n * 2);
    k_fill<<<4096, 256>>>(static_cast<uint32_t*>(b), n * 2);
    CK(hipDeviceSynchronize());
    const unsigned grid = unsigned((n >> 1) / (size_t(kBlock) * 2));
    std::vector<Variant> vs;
#define V2(L, R, OP, NTL, NTR, NAME) vs.push_back({NAME, [=] { k2<L, R, OP, NTL, NTR><<<grid, kBlock>>>(static_cast<const L*>(a), static_cast<const R*>(b), out, n); }, double(sizeof(L) + sizeof(R) + 8), {}})
#define V1(L, NTL, NAME) vs.push_back({NAME, [=] { k1<L, EC_MUL, NTL><<<grid, kBlock>>>(static_cast<const L*>(a), 3.0, out, n); }, double(sizeof(L) + 8), {}})
    V2(uint8_t, uint8_t, EC_ADD, false, false, "u8 + u8    plain  plain ");
    V2(uint8_t, uint8_t, EC_ADD, true, true, "u8 + u8    nt     nt    ");
    V2(uint16_t, uint16_t, EC_ADD, false, false, "u16 + u16  plain  plain ");
    V2(uint16_t, uint16_t, EC_ADD, true, true, "u16 + u16  nt     nt    ");
    V2(uint16_t, uint16_t, EC_ADD, true, false, "u16 + u16  nt     plain ");
    V2(float, float, EC_ADD, false, false, "f32 + f32  plain  plain ");
    V2(float, float, EC_ADD, true, true, "f32 + f32  nt     nt    ");
    V2(float, float, EC_ADD, true, false, "f32 + f32  nt     plain ");
    V2(double, double, EC_ADD, false, false, "f64 + f64  plain  plain ");
    V2(double, double, EC_ADD, true, true, "f64 + f64  nt     nt    ");
    V2(double, double, EC_ADD, true, false, "f64 + f64  nt     plain ");
    V2(uint8_t, uint16_t, EC_DIV, false, false, "u8 / u16   plain  plain ");
    V2(uint8_t, uint16_t, EC_DIV, false, true, "u8 / u16   plain  nt    ");
    V2(uint8_t, uint16_t, EC_DIV, true, false, "u8 / u16   nt     plain ");
    V2(uint8_t, uint16_t, EC_DIV, true, true, "u8 / u16   nt     nt    ");
    V2(uint8_t, double, EC_MUL, false, false, "u8 * f64   plain  plain ");
    V2(uint8_t, double, EC_MUL, false, true, "u8 * f64   plain  nt    ");
    V2(uint16_t, float, EC_SUB, false, false, "u16 - f32  plain  plain ");
    V2(uint16_t, float, EC_SUB, false, true, "u16 - f32  plain  nt    ");
    V2(uint16_t, float, EC_SUB, true, true, "u16 - f32  nt     nt    ");
    V1(uint8_t, false, "u8 * s     plain        ");
    V1(uint8_t, true, "u8 * s     nt           ");
    V1(uint16_t, false, "u16 * s    plain        ");
    V1(uint16_t, true, "u16 * s    nt           ");
    V1(float, false, "f32 * s    plain        ");
    V1(float, true, "f32 * s    nt           ");
    V1(double, false, "f64 * s    plain        ");
    V1(double, true, "f64 * s    nt           ");
    // Is the gain of plain loads on the 1-byte stream a cache effect?  The u8 operand of a 16384² raster is 268 MB — the
    // size of the 256 MiB Infinity Cache — and these loops re-read the same operand every launch.  ROTATING variants read
    // a different 268 MB (u8) / 537 MB (u16) region on each of four consecutive launches, so nothing a launch reads was
    // touched by the three launches before it (1 GB + 2 GB of operands between two uses of a line).
    static int rot = 0;
#define V2ROT(NTL, NTR, NAME) vs.push_back({NAME, [=] { const int k = rot++ & 3; k2<uint8_t, uint16_t, EC_DIV, NTL, NTR><<<grid, kBlock>>>( \
        static_cast<const uint8_t*>(a) + size_t(k) * n, static_cast<const uint16_t*>(b) + size_t(k) * n, out, n); }, 11.0, {}})
#define V1ROT(NTL, NAME) vs.push_back({NAME, [=] { const int k = rot++ & 3; k1<uint8_t, EC_MUL, NTL><<<grid, kBlock>>>(static_cast<const uint8_t*>(a) + size_t(k) * n, 3.0, out, n); }, 9.0, {}})
    V2ROT(false, true, "u8 / u16   plain  nt    ROTATING");
    V2ROT(true, true, "u8 / u16   nt     nt    ROTATING");
    V2ROT(false, false, "u8 / u16   plain  plain ROTATING");
    V1ROT(false, "u8 * s     plain        ROTATING");
    V1ROT(true, "u8 * s     nt           ROTATING");
    const size_t ngroups = n / 2;  // n * 8 bytes of `a` as 16-byte groups
    const unsigned rgrid = 256 * 4;
    vs.push_back({"read 8 B/cell x16 B/lane plain", [=] { k_read<false><<<rgrid, 512>>>(static_cast<const u32x4*>(a), ngroups, sink); }, 8, {}});
    vs.push_back({"read 8 B/cell x16 B/lane nt   ", [=] { k_read<true><<<rgrid, 512>>>(static_cast<const u32x4*>(a), ngroups, sink); }, 8, {}});
    vs.push_back({"read 1 B/cell x16 B/lane plain", [=] { k_read<false><<<rgrid, 512>>>(static_cast<const u32x4*>(a), n / 16, sink); }, 1, {}});
    vs.push_back({"read 1 B/cell x16 B/lane nt   ", [=] { k_read<true><<<rgrid, 512>>>(static_cast<const u32x4*>(a), n / 16, sink); }, 1, {}});
    vs.push_back({"read 1 B/cell plain ROTATING  ", [=] { const int k = rot++ & 7; k_read<false><<<rgrid, 512>>>(static_cast<const u32x4*>(a) + size_t(k) * (n / 16), n / 16, sink); }, 1, {}});
    vs.push_back({"read 1 B/cell nt    ROTATING  ", [=] { const int k = rot++ & 7; k_read<true><<<rgrid, 512>>>(static_cast<const u32x4*>(a) + size_t(k) * (n / 16), n / 16, sink); }, 1, {}});

    for (auto& v : vs) { v.launch(); CK(hipGetLastError()); }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 100; ++i) vs[0].launch();
    std::mt19937 rng(777);
    std::vector<int> order(vs.size());
    for (size_t i = 0; i < vs.size(); ++i) order[i] = int(i);
    for (int r = 0; r < rounds; ++r) {
        std::shuffle(order.begin(), order.end(), rng);
        for (int vi : order) {
            for (int i = 0; i < 5; ++i) vs[vi].launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < 30; ++i) vs[vi].launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            vs[vi].ms.push_back(t / 30);
        }
    }
    printf("%zu x %zu cells, %d interleaved rounds of 30 launches; operand loads of l and r: nt or plain; stores nt\n", side, side, rounds);
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("%-32s median %.4f ms  %.3f of 8 TB/s   (min %.4f ms  %.3f)\n", v.name.c_str(), med, v.bpc * n / (med * 1e-3) / 8e12, mn, v.bpc * n / (mn * 1e-3) / 8e12);
    }
    return 0;
}
