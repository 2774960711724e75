/* An operator tree of any depth in ONE pass from plain C: EVI = 2.5 (nir - red) / (nir + 6 red - 7.5 blue + 1) over three
 * u16 bands, as an expression program (ec_expr) — what the reference evaluates as eight passes with seven f64 temporaries
 * (impl $trt for &CellBuffer, src/buffer.rs:324-352).  The result is compared, bit for bit, with the same eight operators run
 * one by one through ec_binop / ec_binop_scalar, three times: as the library runs it by default (EVI over bands of one width is in
 * its ahead-of-time catalogue: a built-in straight-line kernel), and with that turned off (expr_fixed = 0) interpreted
 * (expr_jit = 0) and compiled for itself through hiprtc (expr_jit = 2).
 *
 *   gcc -std=c99 -Iinclude examples/evi.c -Lerased-cells_amd -lerased_cells_hip -Wl,-rpath,$PWD/erased-cells_amd -o evi
 */
#include <stdio.h>
#include <stdlib.h>
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

enum { N = 100003 }; /* odd on purpose: a tail cell */

static ec_value f64(double x) {
    ec_value v;
    memset(&v, 0, sizeof v);
    v.dtype = EC_F64;
    v.v.f64 = x;
    return v;
}

int main(void) {
    static uint16_t nir[N], red[N], blue[N];
    static double got[N], want[N];
    void *dn = NULL, *dr = NULL, *db = NULL, *out = NULL, *t[7];
    const ec_value k[4] = {f64(2.5), f64(6.0), f64(7.5), f64(1.0)};
    /* r0 = (nir - red) * 2.5;  r1 = nir + red * 6;  r1 = r1 - blue * 7.5;  r1 = r1 + 1;  r0 = r0 / r1 */
    const ec_expr_step prog[8] = {
        {EC_SUB, EC_EXPR_STREAM(0), EC_EXPR_STREAM(1), 0}, {EC_MUL, EC_EXPR_REG(0), EC_EXPR_SCALAR(0), 0},
        {EC_MUL, EC_EXPR_STREAM(1), EC_EXPR_SCALAR(1), 1}, {EC_ADD, EC_EXPR_STREAM(0), EC_EXPR_REG(1), 1},
        {EC_MUL, EC_EXPR_STREAM(2), EC_EXPR_SCALAR(2), 2}, {EC_SUB, EC_EXPR_REG(1), EC_EXPR_REG(2), 1},
        {EC_ADD, EC_EXPR_REG(1), EC_EXPR_SCALAR(3), 1},    {EC_DIV, EC_EXPR_REG(0), EC_EXPR_REG(1), 0}};
    const ec_dtype dt[3] = {EC_U16, EC_U16, EC_U16};
    const void *p[3];
    int64_t interpreted = 0, compiled = 0, builtin = 0;
    size_t i;
    int mode;

    for (i = 0; i < N; ++i) {
        nir[i] = (uint16_t)(9000u + (i * 2654435761u) % 30000u);
        red[i] = (uint16_t)(3000u + (i * 40503u) % 20000u);
        blue[i] = (uint16_t)(1000u + (i * 9973u) % 9000u);
    }
    CHECK(ec_init(0));
    CHECK(ec_alloc(&dn, sizeof nir));
    CHECK(ec_alloc(&dr, sizeof red));
    CHECK(ec_alloc(&db, sizeof blue));
    CHECK(ec_alloc(&out, sizeof got));
    for (i = 0; i < 7; ++i) CHECK(ec_alloc(&t[i], sizeof got));
    CHECK(ec_upload(dn, nir, sizeof nir, NULL));
    CHECK(ec_upload(dr, red, sizeof red, NULL));
    CHECK(ec_upload(db, blue, sizeof blue, NULL));
    /* the reference's evaluation: one operator, one pass, one f64 temporary at a time */
    CHECK(ec_binop(EC_SUB, EC_U16, dn, EC_U16, dr, N, (double *)t[0], NULL));
    CHECK(ec_binop_scalar(EC_MUL, EC_F64, t[0], N, &k[0], (double *)t[1], NULL));
    CHECK(ec_binop_scalar(EC_MUL, EC_U16, dr, N, &k[1], (double *)t[2], NULL));
    CHECK(ec_binop(EC_ADD, EC_U16, dn, EC_F64, t[2], N, (double *)t[3], NULL));
    CHECK(ec_binop_scalar(EC_MUL, EC_U16, db, N, &k[2], (double *)t[4], NULL));
    CHECK(ec_binop(EC_SUB, EC_F64, t[3], EC_F64, t[4], N, (double *)t[5], NULL));
    CHECK(ec_binop_scalar(EC_ADD, EC_F64, t[5], N, &k[3], (double *)t[6], NULL));
    CHECK(ec_binop(EC_DIV, EC_F64, t[1], EC_F64, t[6], N, (double *)out, NULL));
    CHECK(ec_download(want, out, sizeof want, NULL));
    p[0] = dn;
    p[1] = dr;
    p[2] = db;
    for (mode = -1; mode <= 2; mode += mode < 0 ? 1 : 2) { /* the built-in kernel; then, without it: the interpreter kernel, the program compiled for itself (hiprtc) */
        CHECK(ec_tune_set("expr_fixed", mode < 0));
        CHECK(ec_tune_set("expr_jit", mode < 0 ? 0 : mode));
        memset(got, 0, sizeof got);
        CHECK(ec_upload(out, got, sizeof got, NULL));
        CHECK(ec_expr(dt, p, 3, k, 4, prog, 8, N, (double *)out, NULL));
        CHECK(ec_download(got, out, sizeof got, NULL));
        if (memcmp(got, want, sizeof got) != 0) {
            fprintf(stderr, "expr_jit = %d: the one-pass result differs from the eager chain\n", mode);
            return 2;
        }
    }
    CHECK(ec_stat_get("expr_interp_launches", &interpreted));
    CHECK(ec_stat_get("expr_jit_launches", &compiled));
    CHECK(ec_stat_get("expr_fixed_launches", &builtin));
    printf("EVI[0] = %.17g, %d cells, one pass == eight passes; built-in launches %d, interpreted launches %d, compiled launches %d\n", got[0],
           (int)N, (int)builtin, (int)interpreted, (int)compiled);
    CHECK(ec_free(dn));
    CHECK(ec_free(dr));
    CHECK(ec_free(db));
    CHECK(ec_free(out));
    for (i = 0; i < 7; ++i) CHECK(ec_free(t[i]));
    CHECK(ec_shutdown());
    /* the library again after a shutdown: the compiled program is still cached, its module is loaded anew */
    CHECK(ec_init(0));
    CHECK(ec_tune_set("expr_fixed", 0));
    CHECK(ec_tune_set("expr_jit", 2));
    CHECK(ec_alloc(&dn, sizeof nir));
    CHECK(ec_alloc(&dr, sizeof red));
    CHECK(ec_alloc(&db, sizeof blue));
    CHECK(ec_alloc(&out, sizeof got));
    CHECK(ec_upload(dn, nir, sizeof nir, NULL));
    CHECK(ec_upload(dr, red, sizeof red, NULL));
    CHECK(ec_upload(db, blue, sizeof blue, NULL));
    p[0] = dn;
    p[1] = dr;
    p[2] = db;
    memset(got, 0, sizeof got);
    CHECK(ec_expr(dt, p, 3, k, 4, prog, 8, N, (double *)out, NULL));
    CHECK(ec_download(got, out, sizeof got, NULL));
    CHECK(ec_stat_get("expr_jit_launches", &compiled));
    if (memcmp(got, want, sizeof got) != 0 || compiled != 2) {
        fprintf(stderr, "after shutdown + init: result differs or the compiled form did not run (launches %d)\n", (int)compiled);
        return 3;
    }
    CHECK(ec_free(dn));
    CHECK(ec_free(dr));
    CHECK(ec_free(db));
    CHECK(ec_free(out));
    CHECK(ec_shutdown());
    return !(interpreted == 1 && builtin == 1);
}
