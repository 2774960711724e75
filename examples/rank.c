/* The sharded path as ONE PROCESS PER GPU, without torch: every rank owns one row-block of a rows x cols UInt16 raster
 * on its own GPU; `raster / divisor` is local (src/buffer.rs:324-329); BufferOps::min_max (src/buffer.rs:169-173) and
 * Mask::counts (src/masked/mask.rs:72-80) of the whole raster come from two 16-byte all-reduces over xGMI on a
 * communicator the library builds itself.  Rank 0 creates the unique id and leaves it in <uid_file>; the other ranks
 * pick it up there (any other channel — a socket, MPI, an environment variable — does as well).
 *
 *   gcc -std=c99 -D_POSIX_C_SOURCE=200809L -Iinclude examples/rank.c -Lerased-cells_amd -lerased_cells_hip \
 *       -Wl,-rpath,$PWD/erased-cells_amd -o rank
 *   for r in 0 1 2 3 4 5 6 7; do ./rank /tmp/ec.uid 8 $r $r 16384 16384 & done; wait
 *
 * Cells as in examples/sharded.c (x[i] = (i * 2654435761) >> 13 mod 2^16 with 0 / 65535 replaced by 1, then 0 planted in
 * the last shard and 65535 in the first; d[i] = 1 + (i * 40503) % 65535; nodata = 7).  Every rank prints
 *   rank <r>: min <v> max <v> qmin <bits> qmax <bits> data <n> nodata <n>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "erased_cells.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        ec_status st_ = (call);                                                                       \
        if (st_ != EC_OK) {                                                                           \
            fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #call, (int)st_, ec_last_error_string()); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

static int publish_uid(const char *path, const ec_comm_uid *uid) {
    char tmp[1024];
    FILE *f;
    snprintf(tmp, sizeof tmp, "%s.tmp", path);
    f = fopen(tmp, "wb");
    if (!f || fwrite(uid, sizeof *uid, 1, f) != 1) return 1;
    fclose(f);
    return rename(tmp, path); /* atomic: a reader never sees half an id */
}

static int fetch_uid(const char *path, ec_comm_uid *uid) {
    int tries;
    for (tries = 0; tries < 600; ++tries) { /* up to a minute */
        FILE *f = fopen(path, "rb");
        if (f) {
            const size_t got = fread(uid, sizeof *uid, 1, f);
            fclose(f);
            if (got == 1) return 0;
        }
        {
            struct timespec ts;
            ts.tv_sec = 0;
            ts.tv_nsec = 100000000L;
            nanosleep(&ts, NULL);
        }
    }
    return 1;
}

int main(int argc, char **argv) {
    int rank = -1, n_ranks, device;
    uint64_t rows, cols, off, len, k, cells, last_len, first_len, tmp;
    ec_comm_uid uid;
    ec_comm comm = NULL;
    uint16_t *x, *d;
    void *dx = NULL, *dd = NULL, *dq = NULL, *dm = NULL, *payload = NULL;
    ec_value nd, mn, mx, qmn, qmx;
    int64_t keys[2];
    uint64_t counts[2];

    if (argc != 7) {
        fprintf(stderr, "usage: %s <uid_file> <n_ranks> <rank> <device> <rows> <cols>\n", argv[0]);
        return 2;
    }
    n_ranks = atoi(argv[2]);
    rank = atoi(argv[3]);
    device = atoi(argv[4]);
    rows = strtoull(argv[5], NULL, 10);
    cols = strtoull(argv[6], NULL, 10);
    cells = rows * cols;

    CHECK(ec_init(device));
    if (rank == 0) {
        CHECK(ec_comm_get_unique_id(&uid));
        if (publish_uid(argv[1], &uid)) return 3;
    } else if (fetch_uid(argv[1], &uid)) {
        fprintf(stderr, "rank %d: no unique id appeared in %s\n", rank, argv[1]);
        return 3;
    }
    CHECK(ec_comm_init_rank(&uid, n_ranks, rank, &comm)); /* blocks until all ranks have joined */

    /* this rank's row-block, generated from the global cell index (no rank holds the whole raster) */
    CHECK(ec_shard_range(rows, cols, (uint32_t)rank, (uint32_t)n_ranks, &off, &len));
    CHECK(ec_shard_range(rows, cols, (uint32_t)(n_ranks - 1), (uint32_t)n_ranks, &tmp, &last_len));
    CHECK(ec_shard_range(rows, cols, 0, (uint32_t)n_ranks, &tmp, &first_len));
    x = (uint16_t *)malloc((len ? len : 1) * sizeof *x);
    d = (uint16_t *)malloc((len ? len : 1) * sizeof *d);
    if (!x || !d) return 4;
    for (k = 0; k < len; ++k) {
        const uint64_t i = off + k;
        x[k] = (uint16_t)(((i * 2654435761ull) >> 13) & 0xffffu);
        if (x[k] == 0 || x[k] == 65535) x[k] = 1;
        if (cells >= 2 && i == cells - 1 - last_len / 2) x[k] = 0;
        if (cells >= 2 && i == first_len / 2) x[k] = 65535;
        d[k] = (uint16_t)(1 + (i * 40503ull) % 65535ull);
    }

    CHECK(ec_alloc(&dx, len * 2));
    CHECK(ec_alloc(&dd, len * 2));
    CHECK(ec_alloc(&dq, len * 8));
    CHECK(ec_alloc(&dm, len));
    CHECK(ec_alloc(&payload, 32));
    CHECK(ec_upload(dx, x, len * 2, NULL));
    CHECK(ec_upload(dd, d, len * 2, NULL));

    CHECK(ec_binop(EC_DIV, EC_U16, dx, EC_U16, dd, len, (double *)dq, NULL)); /* local: no communication */

    CHECK(ec_min_max_keys(EC_U16, dx, NULL, len, (int64_t *)payload, NULL));
    CHECK(ec_allreduce_min_max_keys(comm, (int64_t *)payload, NULL));
    CHECK(ec_download(keys, payload, sizeof keys, NULL));
    CHECK(ec_min_max_decode(EC_U16, keys, &mn, &mx));

    CHECK(ec_min_max_keys(EC_F64, dq, NULL, len, (int64_t *)payload, NULL));
    CHECK(ec_allreduce_min_max_keys(comm, (int64_t *)payload, NULL));
    CHECK(ec_download(keys, payload, sizeof keys, NULL));
    CHECK(ec_min_max_decode(EC_F64, keys, &qmn, &qmx));

    memset(&nd, 0, sizeof nd);
    nd.dtype = EC_U16;
    nd.v.u16 = 7;
    CHECK(ec_mask_from_nodata(EC_U16, dx, len, &nd, (uint8_t *)dm, NULL));
    CHECK(ec_mask_counts_device((const uint8_t *)dm, len, (uint64_t *)payload, NULL));
    CHECK(ec_allreduce_counts(comm, (uint64_t *)payload, NULL));
    CHECK(ec_download(counts, payload, sizeof counts, NULL));

    printf("rank %d: min %u max %u qmin %llu qmax %llu data %llu nodata %llu\n", rank, (unsigned)mn.v.u16, (unsigned)mx.v.u16,
           (unsigned long long)qmn.v.bits, (unsigned long long)qmx.v.bits, (unsigned long long)counts[0],
           (unsigned long long)counts[1]);

    CHECK(ec_free(dx));
    CHECK(ec_free(dd));
    CHECK(ec_free(dq));
    CHECK(ec_free(dm));
    CHECK(ec_free(payload));
    CHECK(ec_comm_destroy(comm));
    CHECK(ec_shutdown());
    if (rank == 0) remove(argv[1]);
    free(x);
    free(d);
    return 0;
}
