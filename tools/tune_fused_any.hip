// tune_fused_any.hip — k_fused_any (load classes compile-time, kinds and ops launch-uniform: ec_fused_any.hpp) against the
// kernels specialised per cell type and op triple (k_fused_same, k_fused_mixed), on the same buffers, randomised
// interleaved rounds, checksummed outputs (dev tool, round 3).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -Iinclude -Ierased-cells_amd/csrc -Itools \
//         tools/tune_fused_any.hip -o tools/tune_fused_any && ./tools/tune_fused_any [side] [rounds]
//   (-DEC_FUSED_U=N forces one tile depth on every kernel, the legacy ones included)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "ec_fused_any.hpp"
#include "legacy_fused_kernels.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace ecd;

__global__ void k_fill(uint16_t* a, float* b, float* c, float* d, double* e, uint8_t* u8, uint8_t* m0, uint8_t* m1, uint8_t* m2, size_t n) {
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        a[i] = 5000 + splitmix64(7 ^ i) % 35000;
        b[i] = float(5000 + splitmix64(8 ^ i) % 25000);
        c[i] = float(splitmix64(9 ^ i) % 2000) - 1000.f;
        d[i] = float(splitmix64(10 ^ i) % 2000) - 1000.f;
        e[i] = double(splitmix64(14 ^ i) % 4000) - 2000.0;
        u8[i] = uint8_t(splitmix64(15 ^ i));
        m0[i] = splitmix64(11 ^ i) % 100 >= 30;
        m1[i] = splitmix64(12 ^ i) % 100 >= 30;
        m2[i] = splitmix64(13 ^ i) % 100 >= 30;
    }
}

__global__ void k_xor(const uint64_t* p, size_t n, unsigned long long* acc) {
    unsigned long long s = 0;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) s += p[i] * (2 * i + 1);  // bit pattern checksum
    atomicAdd(acc, s);
}

struct Variant {
    std::string name;
    std::function<void()> launch;
    double bpc;
    int group;
    std::vector<float> ms;
};

static FusedArgs make(int nops, const void* const p[4], const int dt[4], int o1, int o2, int o3) {
    FusedArgs fa{};
    fa.o1 = int8_t(o1); fa.o2 = int8_t(o2); fa.o3 = int8_t(o3);
    for (int k = 0; k < 4; ++k) {
        const int src = k < nops ? k : 2;
        fa.p[k] = p[src];
        fa.dt[k] = int8_t(dt[src]);
        fa.alias[k] = int8_t(k);
        for (int j = 0; j < k; ++j)
            if (fa.p[j] == fa.p[k] && fa.dt[j] == fa.dt[k]) { fa.alias[k] = int8_t(j); break; }
    }
    return fa;
}

static size_t size_of_dt(int dt) { static const size_t s[10] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8}; return s[dt]; }

// class bytes of slot k (0: alias / unused)
static int cls(const FusedArgs& fa, int k, int nops) { return (k < nops && fa.alias[k] == k && !fa.is_sc[k]) ? int(size_of_dt(fa.dt[k])) : 0; }

int main(int argc, char** argv) {
    const size_t side = argc > 1 ? strtoull(argv[1], nullptr, 10) : 16384;
    const int rounds = argc > 2 ? atoi(argv[2]) : 9;
    const size_t n = side * side;
    uint16_t* a;
    float *b, *c, *d;
    double* e;
    uint8_t *u8, *m0, *m1, *m2, *om;
    double* out;
    CK(hipMalloc(&a, n * 2)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&e, n * 8));
    CK(hipMalloc(&u8, n)); CK(hipMalloc(&m0, n)); CK(hipMalloc(&m1, n)); CK(hipMalloc(&m2, n)); CK(hipMalloc(&om, n));
    CK(hipMalloc(&out, n * 8));
    k_fill<<<2048, 256>>>(a, b, c, d, e, u8, m0, m1, m2, n);
    CK(hipDeviceSynchronize());
    auto grid_u = [&](int u) { return unsigned((n / 2 + 256 * size_t(u) - 1) / (256 * size_t(u))); };
    auto grid_of = [&](size_t cell_bytes) { return grid_u(fused_u(cell_bytes)); };            // k_fused_any
    auto lgrid_of = [&](size_t cell_bytes) { return grid_u(legacy_fused_u(cell_bytes)); };   // the legacy kernels' own depth

    std::vector<Variant> vs;
    // ---- NDVI u16 (12 B/cell): two u16 streams, z = x, w = y
    {
        const void* p[4] = {a, reinterpret_cast<uint16_t*>(b), a, reinterpret_cast<uint16_t*>(b)};
        const int dt[4] = {EC_U16, EC_U16, EC_U16, EC_U16};
        FusedArgs fa = make(4, p, dt, EC_SUB, EC_DIV, EC_ADD);
        vs.push_back({"NDVI u16            k_fused_same", [=] { k_fused_same<uint16_t, EC_SUB, EC_DIV, EC_ADD><<<lgrid_of(2), 256>>>(fa, out, nullptr, n); }, 12, 0, {}});
        FusedArgs fb = fa; fb.small = 1;
        vs.push_back({"NDVI u16            k_fused_any<2,2,0,0>", [=] { k_fused_any<2, 2, 0, 0><<<grid_of(2), 256>>>(fb, out, nullptr, n); }, 12, 0, {}});
        FusedArgs fc = fa; fc.small = 0;
        vs.push_back({"NDVI u16 (IEEE div) k_fused_any<2,2,0,0>", [=] { k_fused_any<2, 2, 0, 0><<<grid_of(2), 256>>>(fc, out, nullptr, n); }, 12, 0, {}});
    }
    // ---- NDVI u16 + f32 (14 B/cell)
    {
        const void* p[4] = {a, b, a, b};
        const int dt[4] = {EC_U16, EC_F32, EC_U16, EC_F32};
        FusedArgs fa = make(4, p, dt, EC_SUB, EC_DIV, EC_ADD);
        vs.push_back({"NDVI u16+f32        k_fused_mixed", [=] { k_fused_mixed<uint16_t, float, kPatABAB, EC_SUB, EC_DIV, EC_ADD><<<lgrid_of(2), 256>>>(fa, out, nullptr, n); }, 14, 1, {}});
        vs.push_back({"NDVI u16+f32        k_fused_any<2,4,0,0>", [=] { k_fused_any<2, 4, 0, 0><<<grid_of(2), 256>>>(fa, out, nullptr, n); }, 14, 1, {}});
    }
    // ---- config 3: (c + d) * b on f32 with three masks (24 B/cell)
    {
        const void* p[4] = {c, d, b, nullptr};
        const int dt[4] = {EC_F32, EC_F32, EC_F32, EC_F32};
        FusedArgs fa = make(3, p, dt, EC_ADD, EC_MUL, kOpNone);
        fa.m[0] = m0; fa.m[1] = m1; fa.m[2] = m2; fa.nmask = 3;
        vs.push_back({"config 3 f32+masks  k_fused_same", [=] { k_fused_same<float, EC_ADD, EC_MUL, kOpNone><<<lgrid_of(4), 256>>>(fa, out, om, n); }, 24, 2, {}});
        vs.push_back({"config 3 f32+masks  k_fused_any<4,4,4,0>", [=] { k_fused_any<4, 4, 4, 0><<<grid_of(4), 256>>>(fa, out, om, n); }, 24, 2, {}});
    }
    // ---- (c + d) * e : f32, f32, f64 (24 B/cell)
    {
        const void* p[4] = {c, d, e, nullptr};
        const int dt[4] = {EC_F32, EC_F32, EC_F64, EC_F64};
        FusedArgs fa = make(3, p, dt, EC_ADD, EC_MUL, kOpNone);
        vs.push_back({"(f32+f32)*f64       k_fused_mixed AAB", [=] { k_fused_mixed<float, double, kPatAAB, EC_ADD, EC_MUL, kOpNone><<<lgrid_of(4), 256>>>(fa, out, nullptr, n); }, 24, 3, {}});
        vs.push_back({"(f32+f32)*f64       k_fused_any<4,4,8,0>", [=] { k_fused_any<4, 4, 8, 0><<<grid_of(4), 256>>>(fa, out, nullptr, n); }, 24, 3, {}});
    }
    // ---- mixes rounds 1-2 ran as convert-then-fuse
    {
        const void* p[4] = {a, e, a, e};  // NDVI u16 + f64: 18 B/cell
        const int dt[4] = {EC_U16, EC_F64, EC_U16, EC_F64};
        FusedArgs fa = make(4, p, dt, EC_SUB, EC_DIV, EC_ADD);
        vs.push_back({"NDVI u16+f64        k_fused_any<2,8,0,0>", [=] { k_fused_any<2, 8, 0, 0><<<grid_of(2), 256>>>(fa, out, nullptr, n); }, 18, 4, {}});
    }
    {
        const void* p[4] = {u8, a, b, e};  // four cell types: (u8 - u16) / (f32 + f64): 23 B/cell
        const int dt[4] = {EC_U8, EC_U16, EC_F32, EC_F64};
        FusedArgs fa = make(4, p, dt, EC_SUB, EC_DIV, EC_ADD);
        vs.push_back({"(u8-u16)/(f32+f64)  k_fused_any<1,2,4,8>", [=] { k_fused_any<1, 2, 4, 8><<<grid_of(1), 256>>>(fa, out, nullptr, n); }, 23, 5, {}});
    }
    {
        const void* p[4] = {a, a, b, b};  // A A B B: (u16 * u16) + (f32 * f32) on two streams: 14 B/cell
        const int dt[4] = {EC_U16, EC_U16, EC_F32, EC_F32};
        FusedArgs fa = make(4, p, dt, EC_MUL, EC_ADD, EC_MUL);
        vs.push_back({"u16*u16+f32*f32     k_fused_any<2,0,4,0>", [=] { k_fused_any<2, 0, 4, 0><<<grid_of(2), 256>>>(fa, out, nullptr, n); }, 14, 6, {}});
    }

    // outputs of the variants of one group agree bit for bit
    unsigned long long* acc;
    CK(hipMalloc(&acc, 8));
    unsigned long long ref = 0;
    for (size_t i = 0; i < vs.size(); ++i) {
        CK(hipMemset(out, 0xff, n * 8));
        vs[i].launch();
        CK(hipGetLastError());
        CK(hipMemset(acc, 0, 8));
        k_xor<<<2048, 256>>>(reinterpret_cast<const uint64_t*>(out), n, acc);
        unsigned long long h;
        CK(hipMemcpy(&h, acc, 8, hipMemcpyDeviceToHost));
        const bool first = i == 0 || vs[i].group != vs[i - 1].group;
        if (first) ref = h;
        printf("checksum %-44s %016llx%s\n", vs[i].name.c_str(), h, h == ref ? "" : "   <-- DIFFERS from its group's first");
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 100; ++i) vs[0].launch();  // clock ramp
    std::mt19937 rng(4242);
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
#ifdef EC_FUSED_U
    printf("\nbuilt with -DEC_FUSED_U=%d (one tile depth for every kernel)\n", EC_FUSED_U);
#endif
    printf("\n%zu x %zu cells, %d interleaved rounds of 30 launches; median (min)\n", side, side, rounds);
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("%-44s %.4f ms  %6.1f Gcells/s  %.3f of 8 TB/s   (min %.4f ms  %.3f)\n", v.name.c_str(), med, n / (med * 1e-3) / 1e9,
               v.bpc * n / (med * 1e-3) / 8e12, mn, v.bpc * n / (mn * 1e-3) / 8e12);
    }
    (void)cls;
    return 0;
}
