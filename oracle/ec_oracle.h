/*
 * ec_oracle.h — CPU restatement of the erased-cells per-cell arithmetic path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (liberased_cells_hip.so) never links, loads or calls anything here.
 *
 * PINNING: the Rust reference cannot be built in this environment (no
 * rustc/cargo).  The oracle is pinned by the reference's own known-answer
 * tests and fixtures (tests/test_oracle_kat.py restates
 * every one of them, tests/golden/ holds the three Landsat TIFF fixtures of
 * testkit/data and the NDVI numbers quoted at src/gdal/rasterband.rs:150-160).
 * Behaviour no reference test covers (u64/i64 -> f64 rounding above 2^53,
 * NaN/inf arithmetic) is "parity by language spec (Rust `as` casts, IEEE-754),
 * unpinned by reference tests".
 *
 * Third-party arithmetic restated here: num-traits 0.2.17 `ToPrimitive`
 * (Cargo.lock:105-113) — int->int range-checked casts, int->float `as`
 * (round-to-nearest-even), float->int range-checked truncation,
 * float->float `as`.
 *
 * Two forms of every buffer operation:
 *   eco_*   "reference-shaped": 16-byte tagged scalar per cell, per-cell
 *           union/convert dispatch and the two-pass collect of
 *           src/buffer.rs:229-250.  This is what the reference's CPU does and
 *           is the timed cpu_baseline (kind "port", 1 thread).
 *   ecof_*  "typed loops": the same results from monomorphic loops; fast
 *           checker for large inputs (optionally OpenMP-parallel).
 *
 * NaN note: generated NaNs (0/0, inf-inf, 0*inf) carry the host FPU's default
 * NaN.  On x86-64 (the reference's CI platform, .github/workflows/CI.yml
 * `runs-on: ubuntu-latest`) that is 0xFFF8000000000000.  The HIP path
 * canonicalises to the same pattern; see DESIGN.md "NaN policy".
 */
#ifndef EC_ORACLE_H
#define EC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CellType discriminants: order of with_ct! (src/lib.rs:85-101), #[repr(u8)]
 * (src/ctype.rs:16). */
enum {
    ECO_U8 = 0, ECO_U16 = 1, ECO_U32 = 2, ECO_U64 = 3,
    ECO_I8 = 4, ECO_I16 = 5, ECO_I32 = 6, ECO_I64 = 7,
    ECO_F32 = 8, ECO_F64 = 9, ECO_NTYPES = 10
};

enum { ECO_ADD = 0, ECO_SUB = 1, ECO_MUL = 2, ECO_DIV = 3 };

enum { ECO_OK = 0, ECO_ERR_NARROWING = 1, ECO_ERR_BADTYPE = 2, ECO_ERR_NOMEM = 3 };

/* NoData<T> (src/masked/nodata.rs:7-17). */
enum { ECO_ND_NONE = 0, ECO_ND_DEFAULT = 1, ECO_ND_VALUE = 2 };

/* CellValue (src/value.rs:12-20): tag + payload, 16 bytes like the Rust enum. */
typedef struct eco_value {
    uint8_t ct;
    uint8_t pad_[7];
    union {
        uint8_t u8; uint16_t u16; uint32_t u32; uint64_t u64;
        int8_t i8; int16_t i16; int32_t i32; int64_t i64;
        float f32; double f64;
        uint64_t bits;
    } v;
} eco_value;

/* ---- type lattice: src/ctype.rs ---- */
int eco_is_integral(int ct);                 /* ctype.rs:55-68  */
int eco_is_signed(int ct);                   /* ctype.rs:71-84  */
size_t eco_size_of(int ct);                  /* ctype.rs:87-96  */
int eco_union(int a, int b);                 /* ctype.rs:99-126 */
int eco_can_fit_into(int src, int dst);      /* ctype.rs:129-131 */
eco_value eco_min_value(int ct);             /* ctype.rs:158-167 */
eco_value eco_max_value(int ct);             /* ctype.rs:170-179 */

/* ---- scalar: src/value.rs ---- */
int eco_value_convert(const eco_value *v, int ct, eco_value *out); /* value.rs:74-98 */
void eco_value_unify(const eco_value *a, const eco_value *b,
                     eco_value *ua, eco_value *ub);                /* value.rs:103-107 */
eco_value eco_value_binop(int op, const eco_value *l, const eco_value *r); /* value.rs:199-217 */
eco_value eco_value_neg(const eco_value *v);                       /* value.rs:224-240 */
int eco_value_cmp(const eco_value *a, const eco_value *b);         /* value.rs:248-265: -1/0/1 */
int eco_value_eq(const eco_value *a, const eco_value *b);          /* value.rs:267-271 */
double eco_value_to_f64(const eco_value *v);                       /* value.rs:145-156 */

/* NoData::value()/is() (src/masked/nodata.rs:23-49); returns 0 if no value. */
int eco_nodata_value(int kind, int ct, const eco_value *given, eco_value *out);

/* ---- reference-shaped buffer ops ----
 * Outputs are caller-allocated with capacity for the result; out_ct and out_len
 * receive the result cell type and length (empty results are UInt8 per
 * src/buffer.rs:233-234). */
int eco_binop(int op, int lt, const void *l, size_t nl, int rt, const void *r, size_t nr,
              void *out, int *out_ct, size_t *out_len);          /* buffer.rs:324-329 */
int eco_binop_scalar(int op, int lt, const void *l, size_t n, const eco_value *rhs,
                     void *out, int *out_ct, size_t *out_len);   /* buffer.rs:346-352 */
int eco_neg(int t, const void *in, size_t n, void *out, int *out_ct, size_t *out_len); /* buffer.rs:360-365 */
int eco_convert(int st, const void *src, size_t n, int dt, void *dst,
                int *out_ct, size_t *out_len);                   /* buffer.rs:150-167 */
int eco_min_max(int t, const void *p, const uint8_t *mask_or_null, size_t n,
                eco_value *mn, eco_value *mx);  /* buffer.rs:169-173, masked_buffer.rs:208-217 */
int eco_mask_from_nodata(int t, const void *p, size_t n, int nd_kind, const eco_value *nd,
                         uint8_t *mask);                          /* masked_buffer.rs:62-71 */
int eco_mask_select(int t, const void *p, const uint8_t *mask, size_t n, int nd_kind,
                    const eco_value *nd, void *out);              /* masked_buffer.rs:143-151 */
void eco_mask_and(const uint8_t *l, size_t nl, const uint8_t *r, size_t nr, uint8_t *out, size_t *out_len); /* mask.rs:129-140 */
void eco_mask_or(const uint8_t *l, size_t nl, const uint8_t *r, size_t nr, uint8_t *out, size_t *out_len);  /* mask.rs:153-163 */
void eco_mask_not(const uint8_t *m, size_t n, uint8_t *out);     /* mask.rs:111-116 */
void eco_mask_counts(const uint8_t *m, size_t n, uint64_t *n_true, uint64_t *n_false); /* mask.rs:72-80 */
int eco_mask_all(const uint8_t *m, size_t n, int value);         /* mask.rs:67-69 */
/* CellBuffer Ord (buffer.rs:389-436): -1/0/1. */
int eco_buffer_cmp(int lt, const void *l, size_t nl, int rt, const void *r, size_t nr);

/* ---- typed-loop forms (same results; threads = 0 -> serial) ---- */
void ecof_set_threads(int threads);
int ecof_binop(int op, int lt, const void *l, int rt, const void *r, size_t n, double *out);
int ecof_binop_scalar(int op, int lt, const void *l, size_t n, const eco_value *rhs, double *out);
int ecof_neg(int t, const void *in, size_t n, void *out, int *out_ct);
int ecof_convert(int st, const void *src, size_t n, int dt, void *dst);
int ecof_min_max(int t, const void *p, const uint8_t *mask_or_null, size_t n,
                 eco_value *mn, eco_value *mx);
int ecof_mask_from_nodata(int t, const void *p, size_t n, const eco_value *nd_or_null, uint8_t *mask);
int ecof_mask_select(int t, const void *p, const uint8_t *mask, size_t n,
                     const eco_value *nd_or_null, void *out);

/* Synthetic inputs of SURVEY.md §8(d): x[i] = lo + splitmix64(seed ^ i) % (hi-lo+1). */
uint64_t eco_splitmix64(uint64_t x);
void eco_fill_u8(uint8_t *p, size_t n, uint64_t seed, uint64_t base, uint32_t lo, uint32_t hi);
void eco_fill_u16(uint16_t *p, size_t n, uint64_t seed, uint64_t base, uint32_t lo, uint32_t hi);

#ifdef __cplusplus
}
#endif
#endif
