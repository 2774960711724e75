// tune_nt_u8.hip — cache policy of the 1-byte operand stream of k_binop_direct (dev tool, round 3).
//
// `__builtin_nontemporal_load` of an UNDER-ALIGNED 2-byte value (the u8x2 / i8x2 pair of a lane) compiles to a plain
// `global_load_ushort` on gfx950 / ROCm 7.2 — the legaliser drops the non-temporal flag for that one width — so the u8
// stream of the headline kernel (268 MB per launch) was never loaded `nt` while every other stream was.  Forms that
// keep the modifier:
//   P0  under-aligned vector load (what rounds 1-2 shipped: no `nt` in the ISA)
//   P1  under-aligned SCALAR u16 load, the two cells split off with shifts -> global_load_ushort … nt  (any address: it is
//       the <2 x i8> vector type, not the alignment, that loses the flag)
//   P2  raw buffer load with the nt cache bit -> buffer_load_ushort … offen nt  (any address; one descriptor per tile)
// Randomised interleaved rounds; prints the median per variant.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -Iinclude -Ierased-cells_amd/csrc \
//         tools/tune_nt_u8.hip -o tools/tune_nt_u8 && ./tools/tune_nt_u8
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <type_traits>
#include <vector>

#include "ec_binop_kernels.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace ecd;

typedef uint16_t u16_ua __attribute__((aligned(1)));

// A lane's pair of cells as loaded: the typed 2-vector, or for 1-byte cells under P1/P2 the raw 16-bit word (a <2 x i8>
// value anywhere between the load and its use lets the optimiser fold the word back into a vector load, which loses nt).
template <int P, typename T>
struct Pair {
    static constexpr bool kWord = sizeof(T) == 1 && P != 0;
    using raw = typename std::conditional<kWord, uint16_t, vec<T, 2>>::type;
    static __device__ __forceinline__ raw load(const T* tile_base, unsigned pair_in_tile) {
        if constexpr (!kWord) {
            return nt_load(reinterpret_cast<const vec<T, 2>*>(tile_base) + pair_in_tile);
        } else if constexpr (P == 1) {
            return __builtin_nontemporal_load(reinterpret_cast<const u16_ua*>(tile_base) + pair_in_tile);
        } else {
            // wave-uniform descriptor over this tile: base = tile_base, 2^31 records (bounds are the tile's own guard)
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(tile_base), 0, 0x7fffffff, 0x00020000);
            return __builtin_amdgcn_raw_buffer_load_b16(rs, pair_in_tile * 2u, 0, /*aux: nt*/ 2);
        }
    }
    static __device__ __forceinline__ double x(raw v) {
        if constexpr (kWord) return to_f64(static_cast<T>(v & 0xffu));
        else return to_f64(v.x);
    }
    static __device__ __forceinline__ double y(raw v) {
        if constexpr (kWord) return to_f64(static_cast<T>(v >> 8));
        else return to_f64(v.y);
    }
};

template <typename L, typename R, int OP, int U, int P>
__global__ __launch_bounds__(kBlock) void k_try(const L* __restrict__ l, const R* __restrict__ r, double* __restrict__ out, size_t n) {
    using D2 = vec<double, 2>;
    constexpr bool FP = is_fp<L>::value || is_fp<R>::value;
    constexpr bool SM = is_small_int<L>::value && is_small_int<R>::value;
    constexpr size_t TILE = size_t(kBlock) * U;
    const size_t npairs = n >> 1;
    const size_t tile = two_front_tile();
    if (tile * TILE + TILE > npairs) return;  // the tool runs whole tiles only
    const L* lt = l + tile * TILE * 2;
    const R* rt = r + tile * TILE * 2;
    D2* op = reinterpret_cast<D2*>(out) + tile * TILE;
    typename Pair<P, L>::raw a[U];
    typename Pair<P, R>::raw b[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        a[j] = Pair<P, L>::load(lt, threadIdx.x + j * kBlock);
        b[j] = Pair<P, R>::load(rt, threadIdx.x + j * kBlock);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
        D2 o;
        o.x = cell_op<OP, FP, SM>(Pair<P, L>::x(a[j]), Pair<P, R>::x(b[j]));
        o.y = cell_op<OP, FP, SM>(Pair<P, L>::y(a[j]), Pair<P, R>::y(b[j]));
        nt_store(o, op + threadIdx.x + j * kBlock);
    }
}

template <typename L, int OP, int U, int P>
__global__ __launch_bounds__(kBlock) void k_try_scalar(const L* __restrict__ l, double s, double* __restrict__ out, size_t n) {
    using D2 = vec<double, 2>;
    constexpr size_t TILE = size_t(kBlock) * U;
    const size_t npairs = n >> 1;
    const size_t tile = two_front_tile();
    if (tile * TILE + TILE > npairs) return;
    const L* lt = l + tile * TILE * 2;
    D2* op = reinterpret_cast<D2*>(out) + tile * TILE;
    typename Pair<P, L>::raw a[U];
#pragma unroll
    for (int j = 0; j < U; ++j) a[j] = Pair<P, L>::load(lt, threadIdx.x + j * kBlock);
#pragma unroll
    for (int j = 0; j < U; ++j) {
        D2 o;
        o.x = cell_op<OP, true>(Pair<P, L>::x(a[j]), s);
        o.y = cell_op<OP, true>(Pair<P, L>::y(a[j]), s);
        nt_store(o, op + threadIdx.x + j * kBlock);
    }
}

__global__ void k_fill(uint32_t* p, size_t nwords) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        uint32_t w = uint32_t(splitmix64(i));
        p[i] = w | 0x01010101u;  // no zero cell of any width: no division by zero
    }
}

__global__ void k_sum(const double* p, size_t n, double* acc) {
    double s = 0;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) s += p[i];
    atomicAdd(acc, s);
}

struct Variant {
    const char* name;
    void (*launch)();
    double bytes_per_cell;
    std::vector<float> ms;
};

static const void *ga, *gb;
static double* gout;
static size_t gn;

template <typename L, typename R, int OP, int P>
static void L2v() {
    const size_t tiles = (gn >> 1) / (size_t(kBlock) * 2);
    k_try<L, R, OP, 2, P><<<unsigned(tiles), kBlock>>>(static_cast<const L*>(ga), static_cast<const R*>(gb), gout, gn);
}
template <typename L, int OP, int P>
static void L1v() {
    const size_t tiles = (gn >> 1) / (size_t(kBlock) * 2);
    k_try_scalar<L, OP, 2, P><<<unsigned(tiles), kBlock>>>(static_cast<const L*>(ga), 3.0, gout, gn);
}
template <typename L, typename R, int OP>
static void Lib() {  // the library kernel as shipped, for reference
    const size_t tiles = ((gn >> 1) + size_t(kBlock) * 2 - 1) / (size_t(kBlock) * 2);
    k_binop_direct<L, R, OP, 2, true, true><<<unsigned(tiles), kBlock>>>(static_cast<const L*>(ga), static_cast<const R*>(gb), gout, gn, 0u);
}

static double checksum() {
    double* acc;
    CK(hipMalloc(&acc, 8));
    CK(hipMemset(acc, 0, 8));
    k_sum<<<2048, 256>>>(gout, gn, acc);
    double h;
    CK(hipMemcpy(&h, acc, 8, hipMemcpyDeviceToHost));
    CK(hipFree(acc));
    return h;
}

int main(int argc, char** argv) {
    const size_t side = argc > 1 ? strtoull(argv[1], nullptr, 10) : 16384;
    const int rounds = argc > 2 ? atoi(argv[2]) : 11;
    gn = side * side;
    void *a, *b;
    CK(hipMalloc(&a, gn * 2));
    CK(hipMalloc(&b, gn * 2));
    CK(hipMalloc(&gout, gn * 8));
    k_fill<<<4096, 256>>>(static_cast<uint32_t*>(a), gn / 2);
    k_fill<<<4096, 256>>>(static_cast<uint32_t*>(b), gn / 2);
    CK(hipDeviceSynchronize());
    ga = a;
    gb = b;
    std::vector<Variant> vs = {
        {"u8/u16  library k_binop_direct     ", Lib<uint8_t, uint16_t, EC_DIV>, 11, {}},
        {"u8/u16  P0 under-aligned (no nt)   ", L2v<uint8_t, uint16_t, EC_DIV, 0>, 11, {}},
        {"u8/u16  P1 scalar u16 + shifts nt  ", L2v<uint8_t, uint16_t, EC_DIV, 1>, 11, {}},
        {"u8/u16  P2 buffer_load_ushort nt   ", L2v<uint8_t, uint16_t, EC_DIV, 2>, 11, {}},
        {"u8+u16  P0                         ", L2v<uint8_t, uint16_t, EC_ADD, 0>, 11, {}},
        {"u8+u16  P1                         ", L2v<uint8_t, uint16_t, EC_ADD, 1>, 11, {}},
        {"u8+u16  P2                         ", L2v<uint8_t, uint16_t, EC_ADD, 2>, 11, {}},
        {"u8+u8   P0                         ", L2v<uint8_t, uint8_t, EC_ADD, 0>, 10, {}},
        {"u8+u8   P1                         ", L2v<uint8_t, uint8_t, EC_ADD, 1>, 10, {}},
        {"u8+u8   P2                         ", L2v<uint8_t, uint8_t, EC_ADD, 2>, 10, {}},
        {"i8/u8   P0                         ", L2v<int8_t, uint8_t, EC_DIV, 0>, 10, {}},
        {"i8/u8   P1                         ", L2v<int8_t, uint8_t, EC_DIV, 1>, 10, {}},
        {"u8*s    P0                         ", L1v<uint8_t, EC_MUL, 0>, 9, {}},
        {"u8*s    P1                         ", L1v<uint8_t, EC_MUL, 1>, 9, {}},
        {"u8*s    P2                         ", L1v<uint8_t, EC_MUL, 2>, 9, {}},
    };
    // results agree across policies (same cells, same op)
    double ref = 0;
    for (size_t i = 0; i < vs.size(); ++i) {
        vs[i].launch();
        CK(hipDeviceSynchronize());
        const double c = checksum();
        const bool first_of_group = i == 0 || vs[i].name[4] != vs[i - 1].name[4] || vs[i].name[2] != vs[i - 1].name[2] || vs[i].name[0] != vs[i - 1].name[0];
        if (first_of_group) ref = c;
        printf("checksum %-36s %.17g %s\n", vs[i].name, c, c == ref ? "" : "  <-- DIFFERS from its group's first");
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 80; ++i) vs[0].launch();  // clock ramp
    std::mt19937 rng(12345);
    std::vector<int> order(vs.size());
    for (size_t i = 0; i < vs.size(); ++i) order[i] = int(i);
    for (int r = 0; r < rounds; ++r) {
        std::shuffle(order.begin(), order.end(), rng);
        for (int vi : order) {
            for (int i = 0; i < 5; ++i) vs[vi].launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < 30; ++i) vs[vi].launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            vs[vi].ms.push_back(t / 30);
        }
    }
    printf("\n%zu x %zu cells, %d interleaved rounds of 30 launches, median / min per variant\n", side, side, rounds);
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("%s  median %.4f ms  %.3f of 8 TB/s   (min %.4f ms  %.3f)\n", v.name, med, v.bytes_per_cell * gn / (med * 1e-3) / 8e12, mn,
               v.bytes_per_cell * gn / (mn * 1e-3) / 8e12);
    }
    return 0;
}
