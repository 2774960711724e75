/* group_bench.c — what does a sharded call cost the HOST?  (dev tool, round 3; plain C over the ABI)
 *
 * One process drives G shards through an ec_shard_group (G launch threads).  The question the round-2 review asked: a
 * 1/8 shard of the 16384² divide runs for ≈59 µs — can the host issue sharded calls faster than the devices retire
 * them?  Timed here, on ONE GPU with device 0 listed G times under EC_GROUP_HOST_COMBINE (the shards then share the
 * device, so GPU-side times are NOT what 8 GPUs would give; the HOST side — queue pushes, wake-ups, launches — is the
 * same code path):
 *   async     K back-to-back ec_sharded_binop calls, fire-and-forget (round 3), then one ec_shard_group_sync
 *   blocking  the same under EC_GROUP_BLOCKING_ISSUE (rounds 1-2: every call waits until all threads have issued)
 *   plain     G ec_binop calls per step from THIS thread on the shards' streams (no launch threads at all)
 * For each: host µs per sharded call (clock_gettime around the K calls, before the sync), wall µs per call including
 * the final sync, and the GPU time of shard 0's stream per call (HIP events recorded in queue order).
 *
 *   gcc -std=c99 -D_POSIX_C_SOURCE=200809L -D__HIP_PLATFORM_AMD__ -O2 -Iinclude -I/opt/rocm/include tools/group_bench.c \
 *       -Lerased-cells_amd -lerased_cells_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/erased-cells_amd -o tools/group_bench
 *   ./tools/group_bench <G> <cells_per_shard> <K>
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "erased_cells.h"

#define MAXG 64
#define CHECK(call)                                                                     \
    do {                                                                                \
        ec_status st_ = (call);                                                         \
        if (st_ != EC_OK) {                                                             \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, ec_last_error_string()); \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)
#define HCHECK(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));              \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

struct ev_job { hipEvent_t *ev; };
static ec_status record_ev(int32_t shard, int32_t device, ec_stream stream, void *user) {
    (void)device;
    struct ev_job *j = (struct ev_job *)user;
    return hipEventRecord(j->ev[shard], (hipStream_t)stream) == hipSuccess ? EC_OK : EC_ERR_HIP;
}

struct result { double host_us, wall_us, gpu_us; };

static struct result run_group(int G, size_t n, int K, uint32_t extra_flags, void *const *l, void *const *r, double *const *out) {
    int32_t devs[MAXG];
    size_t ns[MAXG];
    hipEvent_t e0[MAXG], e1[MAXG];
    ec_shard_group *g = NULL;
    struct result res;
    for (int i = 0; i < G; ++i) { devs[i] = 0; ns[i] = n; }
    CHECK(ec_shard_group_create(devs, G, EC_GROUP_HOST_COMBINE | extra_flags, &g));
    for (int i = 0; i < G; ++i) { HCHECK(hipEventCreate(&e0[i])); HCHECK(hipEventCreate(&e1[i])); }
    for (int k = 0; k < 60; ++k) CHECK(ec_sharded_binop(g, EC_DIV, EC_U8, (const void *const *)l, EC_U16, (const void *const *)r, ns, out));
    CHECK(ec_shard_group_sync(g));
    struct ev_job j0 = {e0}, j1 = {e1};
    CHECK(ec_shard_group_foreach(g, record_ev, &j0));
    const double t0 = now_us();
    for (int k = 0; k < K; ++k) CHECK(ec_sharded_binop(g, EC_DIV, EC_U8, (const void *const *)l, EC_U16, (const void *const *)r, ns, out));
    const double t1 = now_us();
    CHECK(ec_shard_group_foreach(g, record_ev, &j1));
    CHECK(ec_shard_group_sync(g));
    const double t2 = now_us();
    float ms = 0;
    HCHECK(hipEventElapsedTime(&ms, e0[0], e1[0]));
    res.host_us = (t1 - t0) / K;
    res.wall_us = (t2 - t0) / K;
    res.gpu_us = ms * 1e3 / K;
    for (int i = 0; i < G; ++i) { hipEventDestroy(e0[i]); hipEventDestroy(e1[i]); }
    CHECK(ec_shard_group_destroy(g));
    return res;
}

static struct result run_plain(int G, size_t n, int K, void *const *l, void *const *r, double *const *out) {
    ec_stream st[MAXG];
    hipEvent_t e0, e1;
    struct result res;
    for (int i = 0; i < G; ++i) CHECK(ec_stream_create(&st[i]));
    HCHECK(hipEventCreate(&e0)); HCHECK(hipEventCreate(&e1));
    for (int k = 0; k < 60; ++k)
        for (int i = 0; i < G; ++i) CHECK(ec_binop(EC_DIV, EC_U8, l[i], EC_U16, r[i], n, out[i], st[i]));
    for (int i = 0; i < G; ++i) CHECK(ec_stream_sync(st[i]));
    HCHECK(hipEventRecord(e0, (hipStream_t)st[0]));
    const double t0 = now_us();
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < G; ++i) CHECK(ec_binop(EC_DIV, EC_U8, l[i], EC_U16, r[i], n, out[i], st[i]));
    const double t1 = now_us();
    HCHECK(hipEventRecord(e1, (hipStream_t)st[0]));
    for (int i = 0; i < G; ++i) CHECK(ec_stream_sync(st[i]));
    const double t2 = now_us();
    float ms = 0;
    HCHECK(hipEventElapsedTime(&ms, e0, e1));
    res.host_us = (t1 - t0) / K;
    res.wall_us = (t2 - t0) / K;
    res.gpu_us = ms * 1e3 / K;
    hipEventDestroy(e0); hipEventDestroy(e1);
    for (int i = 0; i < G; ++i) CHECK(ec_stream_destroy(st[i]));
    return res;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <G 1..%d> <cells_per_shard> <K>\n", argv[0], MAXG); return 2; }
    const int G = atoi(argv[1]);
    const size_t n = strtoull(argv[2], NULL, 10);
    const int K = atoi(argv[3]);
    if (G < 1 || G > MAXG || n < 1 || K < 1) return 2;
    void *l[MAXG], *r[MAXG];
    double *out[MAXG];
    CHECK(ec_init(0));
    unsigned short *ones = (unsigned short *)malloc(n * 2);
    for (size_t i = 0; i < n; ++i) ones[i] = (unsigned short)(1 + (i * 40503u) % 65535u);
    for (int i = 0; i < G; ++i) {
        CHECK(ec_alloc(&l[i], n));
        CHECK(ec_alloc(&r[i], n * 2));
        CHECK(ec_alloc((void **)&out[i], n * 8));
        CHECK(ec_upload(l[i], ones, n, NULL));
        CHECK(ec_upload(r[i], ones, n * 2, NULL));
    }
    free(ones);
    const struct result a = run_group(G, n, K, 0, l, r, out);
    const struct result b = run_group(G, n, K, EC_GROUP_BLOCKING_ISSUE, l, r, out);
    const struct result p = run_plain(G, n, K, l, r, out);
    const struct result a2 = run_group(G, n, K, 0, l, r, out);  /* again, after the others: order effects */
    printf("| %d | %zu | %d | %.1f / %.1f / %.1f | %.1f / %.1f / %.1f | %.1f / %.1f / %.1f | %.1f / %.1f / %.1f |\n", G, n, K,
           a.host_us, a.wall_us, a.gpu_us, b.host_us, b.wall_us, b.gpu_us, p.host_us, p.wall_us, p.gpu_us, a2.host_us, a2.wall_us, a2.gpu_us);
    for (int i = 0; i < G; ++i) { CHECK(ec_free(l[i])); CHECK(ec_free(r[i])); CHECK(ec_free(out[i])); }
    CHECK(ec_shutdown());
    return 0;
}
