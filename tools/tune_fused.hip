// tune_fused.hip — is the run-time op dispatch of k_fused_same visible? (dev tool)
// Compares the library's k_fused_same<uint16_t> on NDVI with a fully compile-time NDVI kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ec_fused_kernels.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace ecd;

template <bool FPCHECK>
__global__ __launch_bounds__(256) void k_ndvi_static(const uint16_t* __restrict__ nir, const uint16_t* __restrict__ red, double* __restrict__ out, size_t n) {
    using T2 = vec<uint16_t, 2>;
    const size_t npairs = n >> 1, tile = two_front_tile(), base = tile * 512 + threadIdx.x;
    T2 a[2], b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { size_t pr = base + j * 256; if (pr < npairs) { a[j] = __builtin_nontemporal_load((const T2*)nir + pr); b[j] = __builtin_nontemporal_load((const T2*)red + pr); } }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        size_t pr = base + j * 256;
        if (pr < npairs) {
            D2 o;
            double x0 = a[j].x, y0 = b[j].x, x1 = a[j].y, y1 = b[j].y;
            o.x = cell_op<EC_DIV, FPCHECK>(cell_op<EC_SUB, FPCHECK>(x0, y0), cell_op<EC_ADD, FPCHECK>(x0, y0));
            o.y = cell_op<EC_DIV, FPCHECK>(cell_op<EC_SUB, FPCHECK>(x1, y1), cell_op<EC_ADD, FPCHECK>(x1, y1));
            __builtin_nontemporal_store(o, (D2*)out + pr);
        }
    }
}
__global__ void k_fill(uint16_t* a, uint16_t* b, size_t n) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) { a[i] = 5000 + splitmix64(7 ^ i) % 35000; b[i] = 5000 + splitmix64(8 ^ i) % 25000; }
}
int main() {
    const size_t n = size_t(16384) * 16384;
    uint16_t *a, *b; double* out;
    CK(hipMalloc(&a, n * 2)); CK(hipMalloc(&b, n * 2)); CK(hipMalloc(&out, n * 8));
    k_fill<<<2048, 256>>>(a, b, n); CK(hipDeviceSynchronize());
    FusedArgs fa{};
    fa.p[0] = a; fa.p[1] = b; fa.p[2] = a; fa.p[3] = b;
    for (int k = 0; k < 4; ++k) fa.dt[k] = EC_U16;
    fa.alias[0] = 0; fa.alias[1] = 1; fa.alias[2] = 0; fa.alias[3] = 1;
    fa.o1 = EC_SUB; fa.o2 = EC_DIV; fa.o3 = EC_ADD;
    const unsigned grid = unsigned((n / 2 + 511) / 512), gridlib = unsigned((n / 2 + 256 * kFusedU - 1) / (256 * kFusedU));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto f) {
        for (int i = 0; i < 80; ++i) f();
        std::vector<float> ms;
        for (int r = 0; r < 7; ++r) { CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t / 20); }
        std::sort(ms.begin(), ms.end());
        printf("%-44s %.4f ms  %.1f Gcells/s  %.0f GB/s (%.1f%%)\n", name, ms[3], n / (ms[3] * 1e-3) / 1e9, 12.0 * n / (ms[3] * 1e-3) / 1e9, 12.0 * n / (ms[3] * 1e-3) / 1e9 / 80);
    };
    run("library k_fused_same<u16,Sub,Div,Add>    ", [&] { k_fused_same<uint16_t, EC_SUB, EC_DIV, EC_ADD><<<gridlib, 256>>>(fa, out, nullptr, n); });
    run("static NDVI, FP NaN checks on every step", [&] { k_ndvi_static<true><<<grid, 256>>>(a, b, out, n); });
    run("static NDVI, integer-input NaN handling", [&] { k_ndvi_static<false><<<grid, 256>>>(a, b, out, n); });
    run("library k_fused_same<u16,Sub,Div,Add>    ", [&] { k_fused_same<uint16_t, EC_SUB, EC_DIV, EC_ADD><<<gridlib, 256>>>(fa, out, nullptr, n); });
    return 0;
}
