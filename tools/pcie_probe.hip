// H2D / D2H rate of page-locked host memory by how it was obtained (hipHostMalloc flags, hipHostRegister of malloc'd pages):
// the question behind ec_host_alloc's choice of flags (profiles/r03/host_pipeline.md).
//   hipcc --offload-arch=gfx950 -O2 tools/pcie_probe.hip -o tools/pcie_probe && ./tools/pcie_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::printf("%s -> %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t bytes = size_t(1) << 30;
    void *dev = nullptr, *dev2 = nullptr;
    CK(hipMalloc(&dev, bytes));
    CK(hipMalloc(&dev2, bytes));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    struct Kind { const char* name; unsigned flags; bool reg; };
    const Kind kinds[] = {{"hipHostMalloc default", hipHostMallocDefault, false},
                          {"hipHostMalloc non-coherent", hipHostMallocNonCoherent, false},
                          {"hipHostMalloc coherent", hipHostMallocCoherent, false},
                          {"hipHostMalloc portable|mapped", hipHostMallocPortable | hipHostMallocMapped, false},
                          {"malloc + hipHostRegister", 0, true}};
    std::printf("| page-locked how | H2D GB/s | D2H GB/s | both at once: H2D / D2H GB/s |\n|---|---:|---:|---:|\n");
    for (const Kind& k : kinds) {
        void *h = nullptr, *h2 = nullptr;
        if (k.reg) {
            h = std::malloc(bytes);
            h2 = std::malloc(bytes);
            std::memset(h, 1, bytes);
            std::memset(h2, 2, bytes);
            CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
            CK(hipHostRegister(h2, bytes, hipHostRegisterDefault));
        } else {
            CK(hipHostMalloc(&h, bytes, k.flags));
            CK(hipHostMalloc(&h2, bytes, k.flags));
            std::memset(h, 1, bytes);
            std::memset(h2, 2, bytes);
        }
        double best[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            double t = now();
            CK(hipMemcpyAsync(dev, h, bytes, hipMemcpyHostToDevice, s1));
            CK(hipStreamSynchronize(s1));
            double r = bytes / (now() - t) / 1e9;
            if (r > best[0]) best[0] = r;
            t = now();
            CK(hipMemcpyAsync(h2, dev2, bytes, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s2));
            r = bytes / (now() - t) / 1e9;
            if (r > best[1]) best[1] = r;
            t = now();
            CK(hipMemcpyAsync(dev, h, bytes, hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync(h2, dev2, bytes, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s1));
            const double t1 = now() - t;
            CK(hipStreamSynchronize(s2));
            const double t2 = now() - t;
            if (bytes / t1 / 1e9 > best[2]) best[2] = bytes / t1 / 1e9;
            if (bytes / t2 / 1e9 > best[3]) best[3] = bytes / t2 / 1e9;
        }
        std::printf("| %s | %.1f | %.1f | %.1f / %.1f |\n", k.name, best[0], best[1], best[2], best[3]);
        if (k.reg) {
            CK(hipHostUnregister(h));
            CK(hipHostUnregister(h2));
            std::free(h);
            std::free(h2);
        } else {
            CK(hipHostFree(h));
            CK(hipHostFree(h2));
        }
    }
    return 0;
}
