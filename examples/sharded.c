/* The sharded path from plain C, one process driving the listed GPUs (no torch, no Python):
 * a rows x cols UInt16 raster is cut into contiguous row-blocks (ec_shard_range), one per device of an
 * ec_shard_group; `raster / divisor` runs on every shard (src/buffer.rs:324-329, no communication), then
 * BufferOps::min_max (src/buffer.rs:169-173) of the raster and of the quotient and Mask::counts
 * (src/masked/mask.rs:72-80) of a nodata mask are reduced across the shards — RCCL all-reduce of the
 * 16-byte payloads over xGMI, or folded on the host with host_combine = 1.
 *
 *   gcc -std=c99 -Iinclude examples/sharded.c -Lerased-cells_amd -lerased_cells_hip \
 *       -Wl,-rpath,$PWD/erased-cells_amd -o sharded
 *   ./sharded <rows> <cols> <host_combine 0|1> <device> [<device> ...]
 *
 * Cells: x[i] = (i * 2654435761) >> 13 (mod 2^16), with 0 planted at one cell of the last shard and 65535 at one
 * cell of the first; divisor d[i] = 1 + (i * 40503) % 65535; nodata = 7 (mask = x != 7).  Prints
 *   min <v> max <v> qmin <bits> qmax <bits> data <n> nodata <n>
 * which tests/test_gpu_sharded_group.py recomputes with the oracle.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "erased_cells.h"

#define MAXG 64
#define CHECK(call)                                                                     \
    do {                                                                                \
        ec_status st_ = (call);                                                         \
        if (st_ != EC_OK) {                                                             \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, ec_last_error_string()); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

struct nodata_job { /* per-shard arguments of the ec_shard_group_foreach fan-out below */
    void *const *cells;
    void *const *masks;
    const size_t *n;
    ec_value nd;
};

static ec_status mask_one_shard(int32_t shard, int32_t device, ec_stream stream, void *user) {
    const struct nodata_job *j = (const struct nodata_job *)user;
    (void)device;
    /* MaskedCellBuffer::from_vec_with_nodata (src/masked/masked_buffer.rs:62-71) on this shard's device */
    return ec_mask_from_nodata(EC_U16, j->cells[shard], j->n[shard], &j->nd, (uint8_t *)j->masks[shard], stream);
}

int main(int argc, char **argv) {
    int32_t devices[MAXG];
    int32_t g_n, i;
    uint64_t rows, cols, cells, k;
    uint32_t flags;
    ec_shard_group *g = NULL;
    uint16_t *x, *d;
    size_t n[MAXG], off2[MAXG], bytes2[MAXG], bytes8[MAXG], bytes1[MAXG];
    void *dx[MAXG], *dd[MAXG], *dq[MAXG], *dm[MAXG];
    ec_value mn, mx, qmn, qmx;
    uint64_t n_data = 0, n_nodata = 0;
    struct nodata_job job;

    if (argc < 5 || argc - 4 > MAXG) {
        fprintf(stderr, "usage: %s <rows> <cols> <host_combine 0|1> <device> [<device> ...]\n", argv[0]);
        return 2;
    }
    rows = strtoull(argv[1], NULL, 10);
    cols = strtoull(argv[2], NULL, 10);
    flags = atoi(argv[3]) ? EC_GROUP_HOST_COMBINE : EC_GROUP_RCCL;
    g_n = argc - 4;
    for (i = 0; i < g_n; ++i) devices[i] = atoi(argv[4 + i]);
    cells = rows * cols;

    x = (uint16_t *)malloc(cells * sizeof *x);
    d = (uint16_t *)malloc(cells * sizeof *d);
    if (!x || !d) return 3;
    for (k = 0; k < cells; ++k) {
        x[k] = (uint16_t)(((k * 2654435761ull) >> 13) & 0xffffu);
        if (x[k] == 0 || x[k] == 65535) x[k] = 1; /* the extremes are planted below */
        d[k] = (uint16_t)(1 + (k * 40503ull) % 65535ull);
    }

    CHECK(ec_shard_group_create(devices, g_n, flags, &g));
    for (i = 0; i < g_n; ++i) {
        uint64_t o, l;
        CHECK(ec_shard_range(rows, cols, (uint32_t)i, (uint32_t)g_n, &o, &l));
        n[i] = (size_t)l;
        off2[i] = (size_t)o * 2;
        bytes2[i] = n[i] * 2;
        bytes8[i] = n[i] * 8;
        bytes1[i] = n[i];
    }
    /* the global extremes live in different shards, so no single shard knows the answer */
    if (cells >= 2) {
        x[cells - 1 - (n[g_n - 1] ? n[g_n - 1] / 2 : 0)] = 0;
        x[n[0] / 2] = 65535;
    }

    CHECK(ec_sharded_alloc(g, bytes2, dx));
    CHECK(ec_sharded_alloc(g, bytes2, dd));
    CHECK(ec_sharded_alloc(g, bytes8, dq));
    CHECK(ec_sharded_alloc(g, bytes1, dm));
    CHECK(ec_sharded_upload(g, dx, x, off2, bytes2));
    CHECK(ec_sharded_upload(g, dd, d, off2, bytes2));

    CHECK(ec_sharded_binop(g, EC_DIV, EC_U16, (const void *const *)dx, EC_U16, (const void *const *)dd, n, (double *const *)dq));
    CHECK(ec_sharded_min_max(g, EC_U16, (const void *const *)dx, NULL, n, &mn, &mx));
    CHECK(ec_sharded_min_max(g, EC_F64, (const void *const *)dq, NULL, n, &qmn, &qmx));

    memset(&job, 0, sizeof job);
    job.cells = dx;
    job.masks = dm;
    job.n = n;
    job.nd.dtype = EC_U16;
    job.nd.v.u16 = 7;
    CHECK(ec_shard_group_foreach(g, mask_one_shard, &job));
    CHECK(ec_sharded_counts(g, (const uint8_t *const *)dm, n, &n_data, &n_nodata));

    printf("min %u max %u qmin %llu qmax %llu data %llu nodata %llu\n", (unsigned)mn.v.u16, (unsigned)mx.v.u16,
           (unsigned long long)qmn.v.bits, (unsigned long long)qmx.v.bits, (unsigned long long)n_data,
           (unsigned long long)n_nodata);

    CHECK(ec_sharded_free(g, dx));
    CHECK(ec_sharded_free(g, dd));
    CHECK(ec_sharded_free(g, dq));
    CHECK(ec_sharded_free(g, dm));
    CHECK(ec_shard_group_destroy(g));
    CHECK(ec_shutdown());
    free(x);
    free(d);
    return 0;
}
