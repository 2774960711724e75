// tune_binop_u.hip — tile depth U of k_binop_direct per operand width (dev tool).  Round 1 swept U for the headline
// pair (u8, u16) only and the library uses U = 2 for every pair; this sweeps U in {1, 2, 4} for wider operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -Iinclude -Ierased-cells_amd/csrc \
//         tools/tune_binop_u.hip -o tools/tune_binop_u && ./tools/tune_binop_u
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ec_binop_kernels.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace ecd;

__global__ void k_fill(uint32_t* p, size_t nwords) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nwords; i += stride)
        p[i] = 0x3f800000u | (uint32_t(splitmix64(i)) & 0x007fffffu);  // floats in [1, 2); as ints: large positive numbers
}

static hipEvent_t e0, e1;

template <typename F>
static float timed(F f) {
    for (int i = 0; i < 60; ++i) f();
    std::vector<float> ms;
    for (int r = 0; r < 7; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 40; ++i) f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t / 40);
    }
    std::sort(ms.begin(), ms.end());
    return ms[3];
}

template <typename L, typename R, int OP, int U>
static void one(const char* name, const void* l, const void* r, double* out, size_t n) {
    const size_t tiles = ((n >> 1) + size_t(kBlock) * U - 1) / (size_t(kBlock) * U);
    const float ms = timed([&] {
        k_binop_direct<L, R, OP, U, true, true><<<unsigned(tiles), kBlock>>>(static_cast<const L*>(l), static_cast<const R*>(r), out, n, 0u);
    });
    const double bpc = sizeof(L) + sizeof(R) + 8;
    printf("%-34s U=%d  %.4f ms  %.1f Gcells/s  %.3f of 8 TB/s\n", name, U, ms, n / (ms * 1e-3) / 1e9, bpc * n / (ms * 1e-3) / 1e9 / 8000);
}

template <typename L, typename R, int OP>
static void sweep(const char* name, const void* l, const void* r, double* out, size_t n) {
    for (int rep = 0; rep < 2; ++rep) {
        one<L, R, OP, 1>(name, l, r, out, n);
        one<L, R, OP, 2>(name, l, r, out, n);
        one<L, R, OP, 4>(name, l, r, out, n);
    }
}

int main() {
    const size_t n = size_t(16384) * 16384;
    void *a, *b;
    double* out;
    CK(hipMalloc(&a, n * 8));
    CK(hipMalloc(&b, n * 8));
    CK(hipMalloc(&out, n * 8));
    k_fill<<<4096, 256>>>(static_cast<uint32_t*>(a), n * 2);
    k_fill<<<4096, 256>>>(static_cast<uint32_t*>(b), n * 2);
    CK(hipDeviceSynchronize());
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    sweep<uint8_t, uint16_t, EC_DIV>("u8 / u16", a, b, out, n);
    sweep<uint16_t, uint16_t, EC_ADD>("u16 + u16", a, b, out, n);
    sweep<float, float, EC_ADD>("f32 + f32", a, b, out, n);
    sweep<float, float, EC_DIV>("f32 / f32", a, b, out, n);
    sweep<double, float, EC_MUL>("f64 * f32", a, b, out, n);
    sweep<int64_t, double, EC_ADD>("i64 + f64", a, b, out, n);
    sweep<uint8_t, uint8_t, EC_ADD>("u8 + u8", a, b, out, n);
    // do same-rate streams collide in the memory system?  the second operand and the output moved by odd amounts
    const size_t offs[] = {0, 256, 4096 + 256, (1 << 16) + 512, (1 << 20) + 4096 + 256, (3 << 20) + 0x1100, (17 << 20) + 0x2300};
    for (size_t ob : offs)
        for (size_t oo : {size_t(0), size_t((5 << 20) + 0x900)}) {
            char name[64];
            snprintf(name, sizeof name, "u16+u16 b+%zu out+%zu", ob, oo);
            one<uint16_t, uint16_t, EC_ADD, 2>(name, a, static_cast<char*>(b) + ob, reinterpret_cast<double*>(reinterpret_cast<char*>(out) + oo), n - (32 << 20));
        }
    for (size_t ob : offs) {
        char name[64];
        snprintf(name, sizeof name, "f32+f32 b+%zu", ob);
        one<float, float, EC_ADD, 2>(name, a, static_cast<char*>(b) + ob, out, n - (32 << 20));
    }
    return 0;
}
