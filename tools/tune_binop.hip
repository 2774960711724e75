// tune_binop.hip — A/B harness for the u8 ÷ u16 -> f64 kernel variants (dev tool).
// Builds against the library's own kernel header, times each variant with HIP
// events in interleaved rounds (guide §5.4 rule 24) and prints one line per
// variant plus pure-memory reference kernels with the same byte mix.
//   hipcc --offload-arch=gfx950 -O3 -I include -I erased-cells_amd/csrc tools/tune_binop.hip -o tools/tune_binop
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "ec_binop_kernels.hpp"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

using namespace ecd;

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

// reference: 16-B copy (reads B bytes, writes B bytes)
__global__ __launch_bounds__(256) void k_copy16(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n16) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) d[i] = s[i];
}
template <bool NT>
__global__ __launch_bounds__(256) void k_write16(u32x4* __restrict__ d, size_t n16) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    u32x4 v = {1, 2, 3, 4};
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) store_vec<NT>(d + i, v);
}
__global__ __launch_bounds__(256) void k_read16(const u32x4* __restrict__ s, size_t n16, uint32_t* sink) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) acc ^= s[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) *sink = 1;
}

struct Variant {
    std::string name;
    std::function<void(int grid)> launch;
    int blocks_per_cu;  // 0 = one tile per block ("all")
    size_t tiles;       // number of block tiles for "all"
    double bytes;       // algorithmic bytes per launch
    std::vector<float> ms;
};

int main(int argc, char** argv) {
    size_t side = argc > 1 ? strtoull(argv[1], 0, 10) : 16384;
    int rounds = argc > 2 ? atoi(argv[2]) : 5;
    int iters = argc > 3 ? atoi(argv[3]) : 10;
    const size_t n = side * side;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s  CUs %d  n %zu cells\n", prop.name, ncu, n);

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
    k_fill<<<ncu * 8, 256>>>(a, b, n);
    CK(hipDeviceSynchronize());

    std::vector<Variant> vs;
    const double bytes11 = 11.0 * double(n);
    auto add = [&](std::string name, int bpc, size_t tiles, double bytes, std::function<void(int)> f) {
        vs.push_back(Variant{name, f, bpc, tiles, bytes, {}});
    };

#define DIRECT(OPN, OPV, U, NTS, NTL, BPC)                                                              \
    add(std::string("direct ") + OPN + " U" #U " nts" #NTS " ntl" #NTL " bpc" #BPC, BPC,               \
        (n / 2 + size_t(256) * U - 1) / (size_t(256) * U), bytes11,                                     \
        [=](int grid) { k_binop_direct<uint8_t, uint16_t, OPV, U, NTS, NTL><<<grid, 256>>>(a, b, out, n); })
#define LDSV(OPN, OPV, U, NTS, NTL, BPC)                                                                \
    add(std::string("lds    ") + OPN + " U" #U " nts" #NTS " ntl" #NTL " bpc" #BPC, BPC,               \
        (n / (128 * size_t(U)) + 3) / 4, bytes11,                                                       \
        [=](int grid) { k_binop_lds<uint8_t, uint16_t, OPV, U, NTS, NTL><<<grid, 256>>>(a, b, out, n); })

    // reference memory kernels
    // copy the first half of `out` (4n bytes) onto its second half: 4n read + 4n written
    add("ref copy16 (4B/cell rd + 4B/cell wr) bpc8", 8, 0, 8.0 * n,
        [=](int grid) { k_copy16<<<grid, 256>>>((const u32x4*)out, (u32x4*)out + n / 4, n / 4); });
    add("ref write16 plain 8B/cell bpc8", 8, 0, 8.0 * n, [=](int grid) { k_write16<false><<<grid, 256>>>((u32x4*)out, n / 2); });
    add("ref write16 nt    8B/cell bpc8", 8, 0, 8.0 * n, [=](int grid) { k_write16<true><<<grid, 256>>>((u32x4*)out, n / 2); });
    add("ref read16 8B/cell bpc8", 8, 0, 8.0 * n, [=](int grid) { k_read16<<<grid, 256>>>((const u32x4*)out, n / 2, sink); });

    DIRECT("div", EC_DIV, 1, false, false, 8);
    DIRECT("div", EC_DIV, 2, false, false, 8);
    DIRECT("div", EC_DIV, 4, false, false, 8);
    DIRECT("div", EC_DIV, 8, false, false, 8);
    DIRECT("div", EC_DIV, 4, true, false, 8);
    DIRECT("div", EC_DIV, 4, true, true, 8);
    DIRECT("div", EC_DIV, 4, false, false, 4);
    DIRECT("div", EC_DIV, 4, false, false, 16);
    DIRECT("div", EC_DIV, 4, false, false, 0);
    DIRECT("div", EC_DIV, 4, true, false, 0);
    DIRECT("div", EC_DIV, 8, true, false, 8);
    DIRECT("div", EC_DIV, 8, true, false, 0);
    DIRECT("add", EC_ADD, 4, false, false, 8);
    DIRECT("add", EC_ADD, 4, true, false, 8);
    LDSV("div", EC_DIV, 8, false, false, 8);
    LDSV("div", EC_DIV, 8, true, false, 8);
    LDSV("div", EC_DIV, 8, true, true, 8);
    LDSV("div", EC_DIV, 8, false, false, 4);
    LDSV("div", EC_DIV, 8, false, false, 16);
    LDSV("div", EC_DIV, 8, false, false, 0);
    LDSV("div", EC_DIV, 8, true, false, 0);
    LDSV("div", EC_DIV, 16, false, false, 8);
    LDSV("div", EC_DIV, 16, true, false, 8);
    LDSV("add", EC_ADD, 8, false, false, 8);
    LDSV("add", EC_ADD, 8, true, false, 8);

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<unsigned long long> sums(vs.size(), 0);
    for (int round = -1; round < rounds; ++round) {  // round -1 = warm-up + checksum
        for (size_t vi = 0; vi < vs.size(); ++vi) {
            Variant& v = vs[vi];
            int grid = v.blocks_per_cu ? ncu * v.blocks_per_cu : int(std::min<size_t>(v.tiles, 0x7fffffff));
            if (round < 0) {
                CK(hipMemset(out, 0xEE, 64));
                v.launch(grid);
                CK(hipGetLastError());
                CK(hipMemset(acc, 0, 8));
                k_checksum<<<ncu * 8, 256>>>((const uint64_t*)out, n, acc);
                CK(hipMemcpy(&sums[vi], acc, 8, hipMemcpyDeviceToHost));
                continue;
            }
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) v.launch(grid);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms / iters);
        }
    }
    printf("%-52s %9s %9s %9s %8s  %s\n", "variant", "med_ms", "min_ms", "GB/s", "%of8TB", "checksum");
    for (size_t vi = 0; vi < vs.size(); ++vi) {
        Variant& v = vs[vi];
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        double gbs = v.bytes / (med * 1e-3) / 1e9;
        printf("%-52s %9.4f %9.4f %9.1f %7.1f%%  %016llx\n", v.name.c_str(), med, mn, gbs, gbs / 80.0, sums[vi]);
    }
    return 0;
}
