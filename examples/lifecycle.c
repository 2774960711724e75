/* Runtime lifecycle through the C ABI: init -> pooled allocations, an operator, a reduction -> shutdown (pool trimmed and
 * destroyed, scratch freed) -> init again -> the same work again -> shutdown.  Also: a second stream with its own
 * reduction scratch, a block freed on another stream than it was last used on (ec_free_ordered), ec_pool_trim.
 *
 *   gcc -std=c99 -Iinclude examples/lifecycle.c -Lerased-cells_amd -lerased_cells_hip -Wl,-rpath,$PWD/erased-cells_amd -o lifecycle
 */
#include <stdio.h>
#include <string.h>

#include "erased_cells.h"

#define N 100003
#define CHECK(call)                                                                     \
    do {                                                                                \
        ec_status st_ = (call);                                                         \
        if (st_ != EC_OK) {                                                             \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, ec_last_error_string()); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static uint16_t cells[N];
static double quotients[N];

static int one_round(int round) {
    void *d = NULL, *q = NULL;
    ec_stream s = NULL;
    ec_value mn, mx, two;
    int64_t allocs0 = 0, allocs1 = 0, devices = 0;
    size_t i;

    CHECK(ec_init(0));
    CHECK(ec_stat_get("devices", &devices));
    if (devices != 1) return 2;
    CHECK(ec_stat_get("pool_allocs", &allocs0));
    CHECK(ec_stream_create(&s));
    CHECK(ec_alloc_async(&d, sizeof cells, NULL));                 /* allocated on the default stream ... */
    CHECK(ec_alloc_async(&q, sizeof quotients, s));
    CHECK(ec_upload(d, cells, sizeof cells, NULL));
    memset(&two, 0, sizeof two);
    two.dtype = EC_F64;
    two.v.f64 = 2.0;
    CHECK(ec_binop_scalar(EC_DIV, EC_U16, d, N, &two, (double *)q, s)); /* ... last used on `s` */
    CHECK(ec_min_max(EC_U16, d, NULL, N, &mn, &mx, s));
    CHECK(ec_download(quotients, q, sizeof quotients, s));
    CHECK(ec_free_ordered(d, NULL, s));                             /* back to the pool on its own stream, behind `s` */
    CHECK(ec_free_async(q, s));
    CHECK(ec_stream_sync(s));
    CHECK(ec_stat_get("pool_allocs", &allocs1));
    CHECK(ec_pool_trim(0));
    CHECK(ec_stream_destroy(s));
    CHECK(ec_shutdown());
    CHECK(ec_stat_get("devices", &devices));
    if (devices != 0 || allocs1 - allocs0 != 2) return 3;
    if (mn.v.u16 != 1 || mx.v.u16 != 65535) return 4;
    for (i = 0; i < N; ++i)
        if (quotients[i] != cells[i] / 2.0) return 5;
    printf("round %d ok\n", round);
    return 0;
}

int main(void) {
    size_t i;
    int r;
    for (i = 0; i < N; ++i) cells[i] = (uint16_t)(1 + (i * 40503u) % 65535u);
    cells[17] = 1;
    cells[N - 5] = 65535;
    for (r = 0; r < 3; ++r) {
        int rc = one_round(r);
        if (rc) {
            fprintf(stderr, "round %d failed (%d)\n", r, rc);
            return rc;
        }
    }
    /* after shutdown every compute entry point refuses loudly */
    if (ec_binop_scalar(EC_DIV, EC_U16, cells, 1, NULL, quotients, NULL) == EC_OK) return 6;
    return 0;
}
