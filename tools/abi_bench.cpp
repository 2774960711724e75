// abi_bench.cpp — times ec_binop (u8 ÷ u16 -> f64, 16384²) through the C ABI from plain C++
// with HIP events: the same measurement bench.py makes, without Python/torch in the process (dev tool).
//   hipcc -O2 -Iinclude tools/abi_bench.cpp -o tools/abi_bench -Lerased-cells_amd -lerased_cells_hip -Wl,-rpath,'$ORIGIN/../erased-cells_amd'
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "erased_cells.h"

#define EC(x) do { ec_status s_ = (x); if (s_ != EC_OK) { fprintf(stderr, "ec error %d: %s\n", s_, ec_last_error_string()); exit(1); } } while (0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const size_t side = argc > 1 ? strtoull(argv[1], 0, 10) : 16384, n = side * side;
    const int steps = argc > 2 ? atoi(argv[2]) : 50, op = argc > 3 ? atoi(argv[3]) : EC_DIV;
    EC(ec_init(0));
    void *a, *b, *out;
    EC(ec_alloc(&a, n));
    EC(ec_alloc(&b, n * 2));
    EC(ec_alloc(&out, n * 8));
    EC(ec_synth_fill(EC_U8, a, n, 0x5EED0001, 0, 0.0, 255.0, nullptr));
    EC(ec_synth_fill(EC_U16, b, n, 0x5EED0002, 0, 1.0, 65535.0, nullptr));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 5; ++i) EC(ec_binop(op, EC_U8, a, EC_U16, b, n, (double*)out, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < steps; ++i) EC(ec_binop(op, EC_U8, a, EC_U16, b, n, (double*)out, nullptr));
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= steps;
        printf("abi_bench op=%d: %.4f ms/launch  %.1f Gcells/s  %.1f GB/s  (%.1f%% of 8 TB/s)\n", op, ms, n / (ms * 1e-3) / 1e9,
               11.0 * n / (ms * 1e-3) / 1e9, 11.0 * n / (ms * 1e-3) / 1e9 / 80.0);
    }
    return 0;
}
