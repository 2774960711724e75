// tune_fused2.hip — tile depth of the fused kernels (dev tool): the library's own kernels, built with -DEC_FUSED_U=N.
//   config 3   (a + b) * c on f32 with three masks (24 B/cell)      k_fused_same<float, Add, Mul, none>
//   NDVI u16   (x - y) / (x + y), aliased operands (12 B/cell)       k_fused_same<uint16_t, Sub, Div, Add>
//   NDVI mixed u16 + f32 bands in one pass (14 B/cell)                k_fused_mixed<uint16_t, float, ABAB, Sub, Div, Add>
//   (add -DEC_FUSED_FAST_TILES for the straight-line tiles of ec_fused_kernels.hpp; without -DEC_FUSED_U the per-type depth)
//   for U in 2 4 8; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -DEC_FUSED_U=$U \
//       -Iinclude -Ierased-cells_amd/csrc tools/tune_fused2.hip -o tools/tune_fused2_u$U; done
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ec_fused_mixed.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace ecd;
#ifdef EC_FUSED_FAST_TILES
#define VARIANT " fast tiles "
#else
#define VARIANT " general tile"
#endif

__global__ void k_fill(uint16_t* a, float* b, float* c, float* d, uint8_t* m0, uint8_t* m1, uint8_t* m2, size_t n) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        a[i] = 5000 + splitmix64(7 ^ i) % 35000;
        b[i] = float(5000 + splitmix64(8 ^ i) % 25000);
        c[i] = float(splitmix64(9 ^ i) % 2000) - 1000.f;
        d[i] = float(splitmix64(10 ^ i) % 2000) - 1000.f;
        m0[i] = splitmix64(11 ^ i) % 100 >= 30;
        m1[i] = splitmix64(12 ^ i) % 100 >= 30;
        m2[i] = splitmix64(13 ^ i) % 100 >= 30;
    }
}

int main() {
    const size_t n = size_t(16384) * 16384;
    uint16_t* a;
    float *b, *c, *d;
    uint8_t *m0, *m1, *m2, *om;
    double* out;
    CK(hipMalloc(&a, n * 2)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4));
    CK(hipMalloc(&m0, n)); CK(hipMalloc(&m1, n)); CK(hipMalloc(&m2, n)); CK(hipMalloc(&om, n));
    CK(hipMalloc(&out, n * 8));
    k_fill<<<2048, 256>>>(a, b, c, d, m0, m1, m2, n);
    CK(hipDeviceSynchronize());
    #ifdef EC_FUSED_U
    constexpr int kFusedU = EC_FUSED_U;  // the forced depth of this build
#else
    constexpr int kFusedU = 0;  // per-kernel depth (fused_u): grids are computed per kernel below
#endif
    auto grid_of = [&](size_t cell_bytes) { const size_t u = kFusedU ? size_t(kFusedU) : size_t(fused_u(cell_bytes)); return unsigned((n / 2 + 256 * u - 1) / (256 * u)); };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, double bpc, auto f) {
        for (int i = 0; i < 100; ++i) f();
        std::vector<float> ms;
        for (int r = 0; r < 7; ++r) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 60; ++i) f();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t / 60);
        }
        std::sort(ms.begin(), ms.end());
        printf("U=%d%s  %-34s %.4f ms  %.1f Gcells/s  %.3f of 8 TB/s\n", kFusedU, VARIANT, name, ms[3], n / (ms[3] * 1e-3) / 1e9, bpc * n / (ms[3] * 1e-3) / 1e9 / 8000);
    };
    FusedArgs c3{};  // (c + d) * b with masks
    c3.p[0] = c; c3.p[1] = d; c3.p[2] = b; c3.p[3] = b;
    for (int k = 0; k < 4; ++k) { c3.dt[k] = EC_F32; c3.alias[k] = int8_t(k); }
    c3.alias[3] = 2;
    c3.o1 = EC_ADD; c3.o2 = EC_MUL; c3.o3 = kOpNone;
    c3.m[0] = m0; c3.m[1] = m1; c3.m[2] = m2; c3.nmask = 3;
    FusedArgs c3u = c3;  // the same chain without masks (21 B/cell... 20: 12 read + 8 written)
    c3u.nmask = 0;
    FusedArgs nd{};
    nd.p[0] = a; nd.p[1] = a; nd.p[2] = a; nd.p[3] = a;
    for (int k = 0; k < 4; ++k) nd.dt[k] = EC_U16;
    nd.alias[0] = 0; nd.alias[1] = 1; nd.alias[2] = 0; nd.alias[3] = 1;
    nd.p[1] = reinterpret_cast<uint16_t*>(b);  // a second u16 stream (the first 2n bytes of b, contents irrelevant for timing)
    nd.p[3] = nd.p[1];
    nd.o1 = EC_SUB; nd.o2 = EC_DIV; nd.o3 = EC_ADD;
    FusedArgs mx{};
    mx.p[0] = a; mx.p[1] = b; mx.p[2] = a; mx.p[3] = b;
    mx.dt[0] = EC_U16; mx.dt[1] = EC_F32; mx.dt[2] = EC_U16; mx.dt[3] = EC_F32;
    mx.alias[0] = 0; mx.alias[1] = 1; mx.alias[2] = 0; mx.alias[3] = 1;
    mx.o1 = EC_SUB; mx.o2 = EC_DIV; mx.o3 = EC_ADD;
    for (int rep = 0; rep < 2; ++rep) {
        run("config 3 (a+b)*c f32 + 3 masks", 24, [&] { k_fused_same<float, EC_ADD, EC_MUL, kOpNone><<<grid_of(4), 256>>>(c3, out, om, n); });
        run("(a+b)*c f32, no masks", 20, [&] { k_fused_same<float, EC_ADD, EC_MUL, kOpNone><<<grid_of(4), 256>>>(c3u, out, nullptr, n); });
        run("NDVI u16", 12, [&] { k_fused_same<uint16_t, EC_SUB, EC_DIV, EC_ADD><<<grid_of(2), 256>>>(nd, out, nullptr, n); });
        run("NDVI u16 + f32, one pass", 14, [&] { k_fused_mixed<uint16_t, float, kPatABAB, EC_SUB, EC_DIV, EC_ADD><<<grid_of(2), 256>>>(mx, out, nullptr, n); });
    }
    return 0;
}
