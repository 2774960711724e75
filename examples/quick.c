/* examples/quick.rs:5-11 of the reference, through the C ABI from plain C:
 *   CellBuffer([1,2,3] u8) / CellBuffer([2,4,6] u16) * 0.5 == [0.25, 0.25, 0.25] f64
 *
 *   gcc -std=c99 -Iinclude examples/quick.c -Lerased-cells_amd -lerased_cells_hip -Wl,-rpath,$PWD/erased-cells_amd -o quick
 */
#include <stdio.h>
#include <string.h>

#include "erased_cells.h"

#define CHECK(call)                                                                     \
    do {                                                                                \
        ec_status st_ = (call);                                                         \
        if (st_ != EC_OK) {                                                             \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, ec_last_error_string()); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

int main(void) {
    const uint8_t a[3] = {1, 2, 3};
    const uint16_t b[3] = {2, 4, 6};
    double r[3] = {0, 0, 0};
    void *da = NULL, *db = NULL, *q = NULL, *out = NULL;
    ec_value half;

    CHECK(ec_init(0));
    CHECK(ec_alloc(&da, sizeof a));
    CHECK(ec_alloc(&db, sizeof b));
    CHECK(ec_alloc(&q, sizeof r));
    CHECK(ec_alloc(&out, sizeof r));
    CHECK(ec_upload(da, a, sizeof a, NULL));
    CHECK(ec_upload(db, b, sizeof b, NULL));
    /* buf1 / buf2: u8 and u16 cells both widen to f64 (src/value.rs:207), result is Float64 */
    CHECK(ec_binop(EC_DIV, EC_U8, da, EC_U16, db, 3, (double *)q, NULL));
    /* ... * 0.5: RHS scalar (src/buffer.rs:346-352) */
    memset(&half, 0, sizeof half);
    half.dtype = EC_F64;
    half.v.f64 = 0.5;
    CHECK(ec_binop_scalar(EC_MUL, EC_F64, q, 3, &half, (double *)out, NULL));
    CHECK(ec_download(r, out, sizeof r, NULL));
    printf("%g %g %g\n", r[0], r[1], r[2]);
    CHECK(ec_free(da));
    CHECK(ec_free(db));
    CHECK(ec_free(q));
    CHECK(ec_free(out));
    CHECK(ec_shutdown());
    return !(r[0] == 0.25 && r[1] == 0.25 && r[2] == 0.25);
}
