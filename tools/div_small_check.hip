// div_small_check.hip — exhaustive check of cheaper f64 divides for small-integer operands (dev tool).
//
// For operands that are integers of at most 16 bits (u8, i8, u16, i16 cells widened to f64) the quotient
// a / b has only 98304² possible inputs, so a shorter instruction sequence than the compiler's IEEE expansion
// (div_scale x2, rcp, 4 fma, mul, fma, div_fmas, div_fixup) can be PROVEN bit-exact by trying them all.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off tools/div_small_check.hip -o tools/div_small_check
//   (add -DLO=-131070 -DHI=131070 for the range of sums and differences of two 16-bit cells: 68,717,903,881 pairs)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NEWTON>
__device__ __forceinline__ double div_small(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);  // v_rcp_f64
#pragma unroll
    for (int k = 0; k < NEWTON; ++k) {
        const double e = __builtin_fma(-b, y, 1.0);
        y = __builtin_fma(y, e, y);
    }
    double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    q = __builtin_fma(r, y, q);
    if (b == 0.0) q = a == 0.0 ? __builtin_bit_cast(double, 0xFFF8000000000000ull) : (a > 0.0 ? __builtin_inf() : -__builtin_inf());
    return q;
}

#ifndef LO
#define LO (-32768)  // default: every i8 / u8 / i16 / u16 cell value (98304 of them)
#define HI 65535     // -DLO=-131070 -DHI=131070: sums and differences of two such cells (the fused NDVI shape)
#endif
constexpr int SPAN = HI - LO + 1;

template <int NEWTON>
__global__ void k_check(unsigned long long* bad, unsigned long long* first_bad) {
    const long long b = LO + (long long)blockIdx.x;
    unsigned long long local = 0;
    for (long long ai = threadIdx.x; ai < SPAN; ai += blockDim.x) {
        const double a = double(LO + ai), bd = double(b);
        const double ref = a / bd;  // the compiler's correctly rounded expansion
        const double got = div_small<NEWTON>(a, bd);
        const bool same = __builtin_bit_cast(uint64_t, ref) == __builtin_bit_cast(uint64_t, got) || (ref != ref && got != got && b != 0);
        if (!same) {
            ++local;
            atomicMin(first_bad, (unsigned long long)((ai << 24) | (unsigned long long)(b - LO)));
        }
    }
    if (local) atomicAdd(bad, local);
}

int main() {
    unsigned long long *bad, *first;
    CK(hipMalloc(&bad, 8));
    CK(hipMalloc(&first, 8));
    for (int newton = 1; newton <= 2; ++newton) {
        CK(hipMemset(bad, 0, 8));
        CK(hipMemset(first, 0xFF, 8));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (newton == 1) k_check<1><<<SPAN, 256>>>(bad, first);
        else k_check<2><<<SPAN, 256>>>(bad, first);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long hb = 0, hf = 0;
        CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hf, first, 8, hipMemcpyDeviceToHost));
        printf("newton steps %d: %llu mismatches of %llu pairs (%.1f ms)", newton, hb, (unsigned long long)SPAN * SPAN, ms);
        if (hb) printf("; first at a=%lld b=%lld", (long long)(hf >> 24) + LO, (long long)(hf & 0xFFFFFF) + LO);
        printf("\n");
    }
    return 0;
}
