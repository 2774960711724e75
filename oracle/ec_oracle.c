/*
 * ec_oracle.c — CPU restatement of the erased-cells per-cell arithmetic path.
 * TEST INFRASTRUCTURE ONLY (see ec_oracle.h).  Every function cites the
 * reference lines (relative to /root/reference) it follows.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off, no fast-math: every
 * operation here must stay a single correctly rounded IEEE-754 operation).
 */
#include "ec_oracle.h"

#include <float.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* with_ct! (src/lib.rs:85-101): one expansion per (discriminant, field, C type). */
#define ECO_WITH_CT(X)            \
    X(ECO_U8, u8, uint8_t)        \
    X(ECO_U16, u16, uint16_t)     \
    X(ECO_U32, u32, uint32_t)     \
    X(ECO_U64, u64, uint64_t)     \
    X(ECO_I8, i8, int8_t)         \
    X(ECO_I16, i16, int16_t)      \
    X(ECO_I32, i32, int32_t)      \
    X(ECO_I64, i64, int64_t)      \
    X(ECO_F32, f32, float)        \
    X(ECO_F64, f64, double)

/* A second spelling of the same list so the two can be nested. */
#define ECO_WITH_CT2(X, A1, A2, A3)       \
    X(A1, A2, A3, ECO_U8, u8, uint8_t)    \
    X(A1, A2, A3, ECO_U16, u16, uint16_t) \
    X(A1, A2, A3, ECO_U32, u32, uint32_t) \
    X(A1, A2, A3, ECO_U64, u64, uint64_t) \
    X(A1, A2, A3, ECO_I8, i8, int8_t)     \
    X(A1, A2, A3, ECO_I16, i16, int16_t)  \
    X(A1, A2, A3, ECO_I32, i32, int32_t)  \
    X(A1, A2, A3, ECO_I64, i64, int64_t)  \
    X(A1, A2, A3, ECO_F32, f32, float)    \
    X(A1, A2, A3, ECO_F64, f64, double)

static const uint32_t F32_NAN_BITS = 0x7fc00000u;            /* Rust f32::NAN */
static const uint64_t F64_NAN_BITS = 0x7ff8000000000000ull;  /* Rust f64::NAN */

/* ------------------------------------------------------------------ */
/* value constructors: CellValue::new / into_cell_value (value.rs:24-33,
 * encoding.rs:32-34) */
#define MK(ID, F, T)                              \
    static inline eco_value mk_##F(T x) {         \
        eco_value r;                              \
        memset(&r, 0, sizeof r);                  \
        r.ct = (uint8_t)ID;                       \
        r.v.F = x;                                \
        return r;                                 \
    }
ECO_WITH_CT(MK)
#undef MK

/* ------------------------------------------------------------------ */
/* type lattice: src/ctype.rs */

int eco_is_integral(int ct) { return ct != ECO_F32 && ct != ECO_F64; } /* ctype.rs:55-68 */

int eco_is_signed(int ct) { /* ctype.rs:71-84: floats count as signed */
    return ct == ECO_I8 || ct == ECO_I16 || ct == ECO_I32 || ct == ECO_I64 ||
           ct == ECO_F32 || ct == ECO_F64;
}

size_t eco_size_of(int ct) { /* ctype.rs:87-96 */
    switch (ct) {
#define SZ(ID, F, T) case ID: return sizeof(T);
        ECO_WITH_CT(SZ)
#undef SZ
    }
    return 0;
}

static size_t zmax(size_t a, size_t b) { return a > b ? a : b; }

int eco_union(int a, int b) { /* ctype.rs:99-126, statement for statement */
    size_t sa = eco_size_of(a), sb = eco_size_of(b), min_bytes;
    int ia = eco_is_integral(a), ib = eco_is_integral(b);
    int ga = eco_is_signed(a), gb = eco_is_signed(b);
    if (ia && !ib) min_bytes = zmax(sb, 2 * sa);
    else if (!ia && ib) min_bytes = zmax(sa, 2 * sb);
    else if (ga && !gb) min_bytes = zmax(sa, 2 * sb);
    else if (!ga && gb) min_bytes = zmax(sb, 2 * sa);
    else min_bytes = zmax(sa, sb);
    int is_signed = ga || gb;
    int integral = ia && ib;
    if (min_bytes == 1 && !is_signed && integral) return ECO_U8;
    if (min_bytes == 1 && is_signed && integral) return ECO_I8;
    if (min_bytes == 2 && !is_signed && integral) return ECO_U16;
    if (min_bytes == 2 && is_signed && integral) return ECO_I16;
    if (min_bytes == 4 && !is_signed && integral) return ECO_U32;
    if (min_bytes == 4 && is_signed && integral) return ECO_I32;
    if (min_bytes == 4 && !integral) return ECO_F32;
    if (min_bytes == 8 && !is_signed && integral) return ECO_U64;
    if (min_bytes == 8 && is_signed && integral) return ECO_I64;
    return ECO_F64;
}

int eco_can_fit_into(int src, int dst) { return eco_union(src, dst) == dst; } /* ctype.rs:129-131 */

eco_value eco_min_value(int ct) { /* ctype.rs:158-167: $p::MIN (floats: -MAX, finite) */
    switch (ct) {
        case ECO_U8: return mk_u8(0);
        case ECO_U16: return mk_u16(0);
        case ECO_U32: return mk_u32(0);
        case ECO_U64: return mk_u64(0);
        case ECO_I8: return mk_i8(INT8_MIN);
        case ECO_I16: return mk_i16(INT16_MIN);
        case ECO_I32: return mk_i32(INT32_MIN);
        case ECO_I64: return mk_i64(INT64_MIN);
        case ECO_F32: return mk_f32(-FLT_MAX);
        default: return mk_f64(-DBL_MAX);
    }
}

eco_value eco_max_value(int ct) { /* ctype.rs:170-179 */
    switch (ct) {
        case ECO_U8: return mk_u8(UINT8_MAX);
        case ECO_U16: return mk_u16(UINT16_MAX);
        case ECO_U32: return mk_u32(UINT32_MAX);
        case ECO_U64: return mk_u64(UINT64_MAX);
        case ECO_I8: return mk_i8(INT8_MAX);
        case ECO_I16: return mk_i16(INT16_MAX);
        case ECO_I32: return mk_i32(INT32_MAX);
        case ECO_I64: return mk_i64(INT64_MAX);
        case ECO_F32: return mk_f32(FLT_MAX);
        default: return mk_f64(DBL_MAX);
    }
}

/* ------------------------------------------------------------------ */
/* num-traits 0.2.17 ToPrimitive for the ten primitives, reached through
 * `impl ToPrimitive for CellValue` (value.rs:118-157), which overrides only
 * to_i64 / to_u64 / to_f64; every other to_<p> is the trait default that
 * funnels through one of those three. Return 1 = Some, 0 = None. */

static int tp_to_i64(const eco_value *v, int64_t *o) {
    switch (v->ct) {
        case ECO_U8: *o = v->v.u8; return 1;
        case ECO_U16: *o = v->v.u16; return 1;
        case ECO_U32: *o = v->v.u32; return 1;
        case ECO_U64: if (v->v.u64 > (uint64_t)INT64_MAX) return 0; *o = (int64_t)v->v.u64; return 1;
        case ECO_I8: *o = v->v.i8; return 1;
        case ECO_I16: *o = v->v.i16; return 1;
        case ECO_I32: *o = v->v.i32; return 1;
        case ECO_I64: *o = v->v.i64; return 1;
        case ECO_F32: { /* float_to_signed_int, f32 narrower than i64: [MIN, MAX+1) */
            float f = v->v.f32;
            if (f >= -9223372036854775808.0f && f < 9223372036854775808.0f) { *o = (int64_t)f; return 1; }
            return 0;
        }
        default: {
            double d = v->v.f64;
            if (d >= -9223372036854775808.0 && d < 9223372036854775808.0) { *o = (int64_t)d; return 1; }
            return 0;
        }
    }
}

static int tp_to_u64(const eco_value *v, uint64_t *o) {
    switch (v->ct) {
        case ECO_U8: *o = v->v.u8; return 1;
        case ECO_U16: *o = v->v.u16; return 1;
        case ECO_U32: *o = v->v.u32; return 1;
        case ECO_U64: *o = v->v.u64; return 1;
        case ECO_I8: if (v->v.i8 < 0) return 0; *o = (uint64_t)v->v.i8; return 1;
        case ECO_I16: if (v->v.i16 < 0) return 0; *o = (uint64_t)v->v.i16; return 1;
        case ECO_I32: if (v->v.i32 < 0) return 0; *o = (uint64_t)v->v.i32; return 1;
        case ECO_I64: if (v->v.i64 < 0) return 0; *o = (uint64_t)v->v.i64; return 1;
        case ECO_F32: { /* float_to_unsigned_int: (-1, MAX+1) */
            float f = v->v.f32;
            if (f > -1.0f && f < 18446744073709551616.0f) { *o = (uint64_t)f; return 1; }
            return 0;
        }
        default: {
            double d = v->v.f64;
            if (d > -1.0 && d < 18446744073709551616.0) { *o = (uint64_t)d; return 1; }
            return 0;
        }
    }
}

/* to_f64 is infallible for primitives: Rust `as f64` (ints: round to nearest
 * even; f32: exact). */
double eco_value_to_f64(const eco_value *v) { /* value.rs:145-156 */
    switch (v->ct) {
#define TOF(ID, F, T) case ID: return (double)v->v.F;
        ECO_WITH_CT(TOF)
#undef TOF
    }
    return 0.0;
}

/* ------------------------------------------------------------------ */
/* CellValue::convert (value.rs:74-98) */
int eco_value_convert(const eco_value *v, int ct, eco_value *out) {
    if (ct < 0 || ct >= ECO_NTYPES || v->ct >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    if (!eco_can_fit_into(v->ct, ct)) return ECO_ERR_NARROWING; /* :79-81 */
    if (ct == v->ct) { *out = *v; return ECO_OK; }              /* :83-85 */
    int64_t s;
    uint64_t u;
    switch (ct) { /* :90-94: self.to_<p>().ok_or_else(err)?.into_cell_value() */
        case ECO_U8: if (!tp_to_u64(v, &u) || u > UINT8_MAX) return ECO_ERR_NARROWING; *out = mk_u8((uint8_t)u); break;
        case ECO_U16: if (!tp_to_u64(v, &u) || u > UINT16_MAX) return ECO_ERR_NARROWING; *out = mk_u16((uint16_t)u); break;
        case ECO_U32: if (!tp_to_u64(v, &u) || u > UINT32_MAX) return ECO_ERR_NARROWING; *out = mk_u32((uint32_t)u); break;
        case ECO_U64: if (!tp_to_u64(v, &u)) return ECO_ERR_NARROWING; *out = mk_u64(u); break;
        case ECO_I8: if (!tp_to_i64(v, &s) || s < INT8_MIN || s > INT8_MAX) return ECO_ERR_NARROWING; *out = mk_i8((int8_t)s); break;
        case ECO_I16: if (!tp_to_i64(v, &s) || s < INT16_MIN || s > INT16_MAX) return ECO_ERR_NARROWING; *out = mk_i16((int16_t)s); break;
        case ECO_I32: if (!tp_to_i64(v, &s) || s < INT32_MIN || s > INT32_MAX) return ECO_ERR_NARROWING; *out = mk_i32((int32_t)s); break;
        case ECO_I64: if (!tp_to_i64(v, &s)) return ECO_ERR_NARROWING; *out = mk_i64(s); break;
        case ECO_F32: *out = mk_f32((float)eco_value_to_f64(v)); break; /* default to_f32 = to_f64 then `as f32` */
        default: *out = mk_f64(eco_value_to_f64(v)); break;
    }
    return ECO_OK;
}

/* CellValue::unify (value.rs:103-107) */
void eco_value_unify(const eco_value *a, const eco_value *b, eco_value *ua, eco_value *ub) {
    int dest = eco_union(a->ct, b->ct);
    /* `unwrap`: union(a, a∪b) == a∪b for all 100 pairs, so never fails */
    if (eco_value_convert(a, dest, ua) != ECO_OK) abort();
    if (eco_value_convert(b, dest, ub) != ECO_OK) abort();
}

/* cv_bin_op! (value.rs:199-217): unify, both to f64, op, always Float64. */
eco_value eco_value_binop(int op, const eco_value *l, const eco_value *r) {
    eco_value ul, ur;
    eco_value_unify(l, r, &ul, &ur);
    double a = eco_value_to_f64(&ul), b = eco_value_to_f64(&ur);
    double res;
    switch (op) {
        case ECO_ADD: res = a + b; break;
        case ECO_SUB: res = a - b; break;
        case ECO_MUL: res = a * b; break;
        default: res = a / b; break;
    }
    return mk_f64(res);
}

/* impl Neg for CellValue (value.rs:224-240). Signed MIN wraps (release build). */
eco_value eco_value_neg(const eco_value *v) {
    switch (v->ct) {
        case ECO_U8: return mk_i16((int16_t)(-(int16_t)v->v.u8));
        case ECO_U16: return mk_i32(-(int32_t)v->v.u16);
        case ECO_U32: return mk_f64(-(double)v->v.u32);
        case ECO_U64: return mk_f64(-(double)v->v.u64);
        case ECO_I8: return mk_i8((int8_t)(0u - (uint8_t)v->v.i8));
        case ECO_I16: return mk_i16((int16_t)(0u - (uint16_t)v->v.i16));
        case ECO_I32: return mk_i32((int32_t)(0u - (uint32_t)v->v.i32));
        case ECO_I64: return mk_i64((int64_t)(0ull - (uint64_t)v->v.i64));
        case ECO_F32: { uint32_t b; float f = v->v.f32; memcpy(&b, &f, 4); b ^= 0x80000000u; memcpy(&f, &b, 4); return mk_f32(f); }
        default: { uint64_t b; double d = v->v.f64; memcpy(&b, &d, 8); b ^= 0x8000000000000000ull; memcpy(&d, &b, 8); return mk_f64(d); }
    }
}

/* f32/f64::total_cmp keys (core::f64::total_cmp). */
static inline int32_t f32_key(float f) { int32_t b; memcpy(&b, &f, 4); return b ^ (int32_t)((uint32_t)(b >> 31) >> 1); }
static inline int64_t f64_key(double d) { int64_t b; memcpy(&b, &d, 8); return b ^ (int64_t)((uint64_t)(b >> 63) >> 1); }

#define CMP3(a, b) (((a) > (b)) - ((a) < (b)))

/* impl Ord for CellValue (value.rs:248-265) */
int eco_value_cmp(const eco_value *a, const eco_value *b) {
    eco_value l, r;
    eco_value_unify(a, b, &l, &r);
    switch (l.ct) {
        case ECO_U8: return CMP3(l.v.u8, r.v.u8);
        case ECO_U16: return CMP3(l.v.u16, r.v.u16);
        case ECO_U32: return CMP3(l.v.u32, r.v.u32);
        case ECO_U64: return CMP3(l.v.u64, r.v.u64);
        case ECO_I8: return CMP3(l.v.i8, r.v.i8);
        case ECO_I16: return CMP3(l.v.i16, r.v.i16);
        case ECO_I32: return CMP3(l.v.i32, r.v.i32);
        case ECO_I64: return CMP3(l.v.i64, r.v.i64);
        case ECO_F32: { int32_t x = f32_key(l.v.f32), y = f32_key(r.v.f32); return CMP3(x, y); }
        default: { int64_t x = f64_key(l.v.f64), y = f64_key(r.v.f64); return CMP3(x, y); }
    }
}

int eco_value_eq(const eco_value *a, const eco_value *b) { return eco_value_cmp(a, b) == 0; } /* value.rs:267-271 */

/* NoData::value (nodata.rs:23-40) */
int eco_nodata_value(int kind, int ct, const eco_value *given, eco_value *out) {
    if (kind == ECO_ND_NONE) return 0;
    if (kind == ECO_ND_VALUE) { *out = *given; return 1; }
    switch (ct) {
        case ECO_F32: { float f; memcpy(&f, &F32_NAN_BITS, 4); *out = mk_f32(f); return 1; }
        case ECO_F64: { double d; memcpy(&d, &F64_NAN_BITS, 8); *out = mk_f64(d); return 1; }
        default: *out = eco_min_value(ct); return 1; /* <int>::MIN */
    }
}

/* ------------------------------------------------------------------ */
/* buffer plumbing */

/* BufferOps::get (buffer.rs:125-134) */
static inline eco_value buf_get(int ct, const void *p, size_t i) {
    switch (ct) {
#define GET(ID, F, T) case ID: return mk_##F(((const T *)p)[i]);
        ECO_WITH_CT(GET)
#undef GET
    }
    abort();
}

/* Vec<CellValue> grown the way `collect()` grows it when the iterator has no
 * size_hint (CellBufferIterator, buffer.rs:293-305): capacity 4, then doubling. */
typedef struct { eco_value *p; size_t len, cap; } vvec;

static int vvec_push(vvec *v, eco_value x) {
    if (v->len == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 4;
        eco_value *np = (eco_value *)realloc(v->p, nc * sizeof(eco_value));
        if (!np) return 0;
        v->p = np;
        v->cap = nc;
    }
    v->p[v->len++] = x;
    return 1;
}

/* impl FromIterator<CellValue> for CellBuffer, second half (buffer.rs:233-248):
 * cell type = first value's; every value goes through get::<T>() =
 * convert + static_cast (value.rs:51-67). Empty -> UInt8 buffer. */
static int collect_values(vvec *vals, void *out, int *out_ct, size_t *out_len) {
    if (vals->len == 0) {
        *out_ct = ECO_U8;
        *out_len = 0;
        free(vals->p);
        return ECO_OK;
    }
    int ct = vals->p[0].ct;
    for (size_t i = 0; i < vals->len; i++) {
        eco_value c;
        if (eco_value_convert(&vals->p[i], ct, &c) != ECO_OK) abort(); /* `.unwrap()` */
        switch (ct) {
#define PUT(ID, F, T) case ID: ((T *)out)[i] = c.v.F; break;
            ECO_WITH_CT(PUT)
#undef PUT
        }
    }
    *out_ct = ct;
    *out_len = vals->len;
    free(vals->p);
    return ECO_OK;
}

/* impl $trt for &CellBuffer (buffer.rs:324-329): zip truncates to the shorter. */
int eco_binop(int op, int lt, const void *l, size_t nl, int rt, const void *r, size_t nr,
              void *out, int *out_ct, size_t *out_len) {
    if (lt < 0 || lt >= ECO_NTYPES || rt < 0 || rt >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    size_t n = nl < nr ? nl : nr;
    vvec vals = {0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        eco_value a = buf_get(lt, l, i), b = buf_get(rt, r, i);
        if (!vvec_push(&vals, eco_value_binop(op, &a, &b))) { free(vals.p); return ECO_ERR_NOMEM; }
    }
    return collect_values(&vals, out, out_ct, out_len);
}

/* impl $trt<R: Into<CellValue>> for CellBuffer (buffer.rs:346-352) */
int eco_binop_scalar(int op, int lt, const void *l, size_t n, const eco_value *rhs,
                     void *out, int *out_ct, size_t *out_len) {
    if (lt < 0 || lt >= ECO_NTYPES || rhs->ct >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    vvec vals = {0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        eco_value a = buf_get(lt, l, i);
        if (!vvec_push(&vals, eco_value_binop(op, &a, rhs))) { free(vals.p); return ECO_ERR_NOMEM; }
    }
    return collect_values(&vals, out, out_ct, out_len);
}

/* impl Neg for &CellBuffer (buffer.rs:360-365) */
int eco_neg(int t, const void *in, size_t n, void *out, int *out_ct, size_t *out_len) {
    if (t < 0 || t >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    vvec vals = {0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        eco_value a = buf_get(t, in, i);
        if (!vvec_push(&vals, eco_value_neg(&a))) { free(vals.p); return ECO_ERR_NOMEM; }
    }
    return collect_values(&vals, out, out_ct, out_len);
}

/* BufferOps::convert for CellBuffer (buffer.rs:150-167) */
int eco_convert(int st, const void *src, size_t n, int dt, void *dst, int *out_ct, size_t *out_len) {
    if (st < 0 || st >= ECO_NTYPES || dt < 0 || dt >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    if (dt == st) { /* :151-153 clone */
        memcpy(dst, src, n * eco_size_of(st));
        *out_ct = st;
        *out_len = n;
        return ECO_OK;
    }
    if (!eco_can_fit_into(st, dt)) return ECO_ERR_NARROWING; /* :157-159, before touching data */
    vvec vals = {0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        eco_value a = buf_get(st, src, i), c;
        if (eco_value_convert(&a, dt, &c) != ECO_OK) abort(); /* :163 unwrap */
        if (!vvec_push(&vals, c)) { free(vals.p); return ECO_ERR_NOMEM; }
    }
    return collect_values(&vals, dst, out_ct, out_len);
}

/* BufferOps::min_max (buffer.rs:169-173; masked: masked_buffer.rs:208-217).
 * Ord::min keeps lhs on ties, Ord::max takes rhs on ties. */
int eco_min_max(int t, const void *p, const uint8_t *mask, size_t n, eco_value *mn, eco_value *mx) {
    if (t < 0 || t >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    eco_value amin = eco_max_value(t), amax = eco_min_value(t);
    for (size_t i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        eco_value v = buf_get(t, p, i);
        if (eco_value_cmp(&amin, &v) > 0) amin = v;
        if (eco_value_cmp(&amax, &v) <= 0) amax = v;
    }
    *mn = amin;
    *mx = amax;
    return ECO_OK;
}

/* MaskedCellBuffer::from_vec_with_nodata (masked_buffer.rs:62-71) +
 * NoData::is (nodata.rs:42-49). */
int eco_mask_from_nodata(int t, const void *p, size_t n, int nd_kind, const eco_value *nd, uint8_t *mask) {
    if (t < 0 || t >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    eco_value ndv;
    int has = eco_nodata_value(nd_kind, t, nd, &ndv);
    for (size_t i = 0; i < n; i++) {
        eco_value v = buf_get(t, p, i);
        int is_nd = has ? eco_value_eq(&ndv, &v) : 0;
        mask[i] = (uint8_t)!is_nd;
    }
    return ECO_OK;
}

/* MaskedCellBuffer::to_vec_with_nodata, after the to_vec::<T> conversion
 * (masked_buffer.rs:143-151). */
int eco_mask_select(int t, const void *p, const uint8_t *mask, size_t n, int nd_kind,
                    const eco_value *nd, void *out) {
    if (t < 0 || t >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    eco_value ndv;
    int has = eco_nodata_value(nd_kind, t, nd, &ndv);
    size_t sz = eco_size_of(t);
    if (!has) { memcpy(out, p, n * sz); return ECO_OK; }
    if (ndv.ct != t) return ECO_ERR_BADTYPE;
    for (size_t i = 0; i < n; i++) {
        const void *srcp = mask[i] ? (const void *)((const char *)p + i * sz) : (const void *)&ndv.v;
        memcpy((char *)out + i * sz, srcp, sz);
    }
    return ECO_OK;
}

/* impl BitAnd for &Mask (mask.rs:129-140): zip -> shorter length. */
void eco_mask_and(const uint8_t *l, size_t nl, const uint8_t *r, size_t nr, uint8_t *out, size_t *out_len) {
    size_t n = nl < nr ? nl : nr;
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)((l[i] != 0) & (r[i] != 0));
    *out_len = n;
}

/* impl BitOr for &Mask (mask.rs:153-163) */
void eco_mask_or(const uint8_t *l, size_t nl, const uint8_t *r, size_t nr, uint8_t *out, size_t *out_len) {
    size_t n = nl < nr ? nl : nr;
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)((l[i] != 0) | (r[i] != 0));
    *out_len = n;
}

/* impl Not for &Mask (mask.rs:111-116) */
void eco_mask_not(const uint8_t *m, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)!m[i];
}

/* Mask::counts (mask.rs:72-80) */
void eco_mask_counts(const uint8_t *m, size_t n, uint64_t *n_true, uint64_t *n_false) {
    uint64_t d = 0, nd = 0;
    for (size_t i = 0; i < n; i++) { if (m[i]) d++; else nd++; }
    *n_true = d;
    *n_false = nd;
}

/* Mask::all (mask.rs:67-69) */
int eco_mask_all(const uint8_t *m, size_t n, int value) {
    for (size_t i = 0; i < n; i++) if ((m[i] != 0) != (value != 0)) return 0;
    return 1;
}

/* impl Ord for CellBuffer (buffer.rs:389-436): cell type first, then
 * lexicographic (total_cmp for floats), then length. */
int eco_buffer_cmp(int lt, const void *l, size_t nl, int rt, const void *r, size_t nr) {
    if (lt != rt) return CMP3(lt, rt);
    size_t n = nl < nr ? nl : nr;
    for (size_t i = 0; i < n; i++) {
        int c;
        switch (lt) {
            case ECO_F32: { int32_t x = f32_key(((const float *)l)[i]), y = f32_key(((const float *)r)[i]); c = CMP3(x, y); break; }
            case ECO_F64: { int64_t x = f64_key(((const double *)l)[i]), y = f64_key(((const double *)r)[i]); c = CMP3(x, y); break; }
#define CI(ID, F, T) case ID: { T x = ((const T *)l)[i], y = ((const T *)r)[i]; c = CMP3(x, y); break; }
            CI(ECO_U8, u8, uint8_t) CI(ECO_U16, u16, uint16_t) CI(ECO_U32, u32, uint32_t) CI(ECO_U64, u64, uint64_t)
            CI(ECO_I8, i8, int8_t) CI(ECO_I16, i16, int16_t) CI(ECO_I32, i32, int32_t) CI(ECO_I64, i64, int64_t)
#undef CI
            default: return 0;
        }
        if (c) return c;
    }
    return CMP3(nl, nr);
}

/* ------------------------------------------------------------------ */
/* typed-loop forms.  binop ≡ (l as f64) op (r as f64) for every type pair
 * (the unify step is value-preserving or is the same RNE
 * rounding `to_f64` performs).  tests/test_oracle_forms.py checks eco_* ==
 * ecof_* on all 100 pairs × 4 ops. */

static int g_threads = 0;
void ecof_set_threads(int threads) { g_threads = threads; }

#if defined(_OPENMP)
#define PAR_FOR _Pragma("omp parallel for schedule(static) if (g_threads > 1) num_threads(g_threads > 1 ? g_threads : 1)")
#else
#define PAR_FOR
#endif

#define DEF_BINOP(LID, LF, LT, RID, RF, RT)                                              \
    static void fb_##LF##_##RF(int op, const LT *l, const RT *r, size_t n, double *o) {  \
        switch (op) {                                                                    \
            case ECO_ADD: PAR_FOR for (size_t i = 0; i < n; i++) o[i] = (double)l[i] + (double)r[i]; break; \
            case ECO_SUB: PAR_FOR for (size_t i = 0; i < n; i++) o[i] = (double)l[i] - (double)r[i]; break; \
            case ECO_MUL: PAR_FOR for (size_t i = 0; i < n; i++) o[i] = (double)l[i] * (double)r[i]; break; \
            default:      PAR_FOR for (size_t i = 0; i < n; i++) o[i] = (double)l[i] / (double)r[i]; break; \
        }                                                                                \
    }
#define DEF_BINOP_ROW(LID, LF, LT) ECO_WITH_CT2(DEF_BINOP, LID, LF, LT)
ECO_WITH_CT(DEF_BINOP_ROW)
#undef DEF_BINOP
#undef DEF_BINOP_ROW

int ecof_binop(int op, int lt, const void *l, int rt, const void *r, size_t n, double *out) {
    if (lt < 0 || lt >= ECO_NTYPES || rt < 0 || rt >= ECO_NTYPES) return ECO_ERR_BADTYPE;
#define CALL(LID, LF, LT, RID, RF, RT) \
    if (lt == LID && rt == RID) { fb_##LF##_##RF(op, (const LT *)l, (const RT *)r, n, out); return ECO_OK; }
#define CALL_ROW(LID, LF, LT) ECO_WITH_CT2(CALL, LID, LF, LT)
    ECO_WITH_CT(CALL_ROW)
#undef CALL
#undef CALL_ROW
    return ECO_ERR_BADTYPE;
}

int ecof_binop_scalar(int op, int lt, const void *l, size_t n, const eco_value *rhs, double *out) {
    if (lt < 0 || lt >= ECO_NTYPES || rhs->ct >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    const double s = eco_value_to_f64(rhs);
    switch (lt) {
#define SC(ID, F, T)                                                                         \
    case ID: {                                                                               \
        const T *a = (const T *)l;                                                           \
        switch (op) {                                                                        \
            case ECO_ADD: PAR_FOR for (size_t i = 0; i < n; i++) out[i] = (double)a[i] + s; break; \
            case ECO_SUB: PAR_FOR for (size_t i = 0; i < n; i++) out[i] = (double)a[i] - s; break; \
            case ECO_MUL: PAR_FOR for (size_t i = 0; i < n; i++) out[i] = (double)a[i] * s; break; \
            default:      PAR_FOR for (size_t i = 0; i < n; i++) out[i] = (double)a[i] / s; break; \
        }                                                                                    \
        return ECO_OK;                                                                       \
    }
        ECO_WITH_CT(SC)
#undef SC
    }
    return ECO_ERR_BADTYPE;
}

int ecof_neg(int t, const void *in, size_t n, void *out, int *out_ct) {
    switch (t) {
        case ECO_U8: { const uint8_t *a = in; int16_t *o = out; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = (int16_t)(-(int16_t)a[i]); *out_ct = ECO_I16; break; }
        case ECO_U16: { const uint16_t *a = in; int32_t *o = out; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = -(int32_t)a[i]; *out_ct = ECO_I32; break; }
        case ECO_U32: { const uint32_t *a = in; double *o = out; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = -(double)a[i]; *out_ct = ECO_F64; break; }
        case ECO_U64: { const uint64_t *a = in; double *o = out; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = -(double)a[i]; *out_ct = ECO_F64; break; }
        case ECO_I8: { const uint8_t *a = in; uint8_t *o = out; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = (uint8_t)(0u - a[i]); *out_ct = ECO_I8; break; }
        case ECO_I16: { const uint16_t *a = in; uint16_t *o = out; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = (uint16_t)(0u - a[i]); *out_ct = ECO_I16; break; }
        case ECO_I32: { const uint32_t *a = in; uint32_t *o = out; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = 0u - a[i]; *out_ct = ECO_I32; break; }
        case ECO_I64: { const uint64_t *a = in; uint64_t *o = out; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = 0ull - a[i]; *out_ct = ECO_I64; break; }
        case ECO_F32: { const uint32_t *a = in; uint32_t *o = out; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = a[i] ^ 0x80000000u; *out_ct = ECO_F32; break; }
        case ECO_F64: { const uint64_t *a = in; uint64_t *o = out; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = a[i] ^ 0x8000000000000000ull; *out_ct = ECO_F64; break; }
        default: return ECO_ERR_BADTYPE;
    }
    if (n == 0) *out_ct = ECO_U8; /* empty collect (buffer.rs:234) */
    return ECO_OK;
}

#define DEF_CONV(SID, SF, ST, DID, DF, DT)                                     \
    static void fc_##SF##_##DF(const ST *s, size_t n, DT *d) {                 \
        PAR_FOR for (size_t i = 0; i < n; i++) d[i] = (DT)s[i];                \
    }
#define DEF_CONV_ROW(SID, SF, ST) ECO_WITH_CT2(DEF_CONV, SID, SF, ST)
ECO_WITH_CT(DEF_CONV_ROW)
#undef DEF_CONV
#undef DEF_CONV_ROW

/* Typed convert: caller keeps the empty -> UInt8 quirk (eco_convert reports it). */
int ecof_convert(int st, const void *src, size_t n, int dt, void *dst) {
    if (st < 0 || st >= ECO_NTYPES || dt < 0 || dt >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    if (!eco_can_fit_into(st, dt)) return ECO_ERR_NARROWING;
#define CALLC(SID, SF, ST, DID, DF, DT) \
    if (st == SID && dt == DID) { fc_##SF##_##DF((const ST *)src, n, (DT *)dst); return ECO_OK; }
#define CALLC_ROW(SID, SF, ST) ECO_WITH_CT2(CALLC, SID, SF, ST)
    ECO_WITH_CT(CALLC_ROW)
#undef CALLC
#undef CALLC_ROW
    return ECO_ERR_BADTYPE;
}

int ecof_min_max(int t, const void *p, const uint8_t *mask, size_t n, eco_value *mn, eco_value *mx) {
    switch (t) {
#define MMI(ID, F, T, LO, HI)                                                   \
    case ID: {                                                                  \
        const T *a = (const T *)p;                                              \
        T lo = HI, hi = LO;                                                     \
        for (size_t i = 0; i < n; i++) {                                        \
            if (mask && !mask[i]) continue;                                     \
            if (a[i] < lo) lo = a[i];                                           \
            if (a[i] >= hi) hi = a[i];                                          \
        }                                                                       \
        *mn = mk_##F(lo);                                                       \
        *mx = mk_##F(hi);                                                       \
        return ECO_OK;                                                          \
    }
        MMI(ECO_U8, u8, uint8_t, 0, UINT8_MAX)
        MMI(ECO_U16, u16, uint16_t, 0, UINT16_MAX)
        MMI(ECO_U32, u32, uint32_t, 0, UINT32_MAX)
        MMI(ECO_U64, u64, uint64_t, 0, UINT64_MAX)
        MMI(ECO_I8, i8, int8_t, INT8_MIN, INT8_MAX)
        MMI(ECO_I16, i16, int16_t, INT16_MIN, INT16_MAX)
        MMI(ECO_I32, i32, int32_t, INT32_MIN, INT32_MAX)
        MMI(ECO_I64, i64, int64_t, INT64_MIN, INT64_MAX)
#undef MMI
        case ECO_F32: {
            const float *a = (const float *)p;
            float lo = FLT_MAX, hi = -FLT_MAX;
            int32_t klo = f32_key(lo), khi = f32_key(hi);
            for (size_t i = 0; i < n; i++) {
                if (mask && !mask[i]) continue;
                int32_t k = f32_key(a[i]);
                if (k < klo) { klo = k; lo = a[i]; }
                if (k >= khi) { khi = k; hi = a[i]; }
            }
            *mn = mk_f32(lo);
            *mx = mk_f32(hi);
            return ECO_OK;
        }
        case ECO_F64: {
            const double *a = (const double *)p;
            double lo = DBL_MAX, hi = -DBL_MAX;
            int64_t klo = f64_key(lo), khi = f64_key(hi);
            for (size_t i = 0; i < n; i++) {
                if (mask && !mask[i]) continue;
                int64_t k = f64_key(a[i]);
                if (k < klo) { klo = k; lo = a[i]; }
                if (k >= khi) { khi = k; hi = a[i]; }
            }
            *mn = mk_f64(lo);
            *mx = mk_f64(hi);
            return ECO_OK;
        }
    }
    return ECO_ERR_BADTYPE;
}

/* Equality under the total order is bit equality for every cell type. */
int ecof_mask_from_nodata(int t, const void *p, size_t n, const eco_value *nd, uint8_t *mask) {
    if (t < 0 || t >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    if (!nd) { memset(mask, 1, n); return ECO_OK; }
    if (nd->ct != t) return ECO_ERR_BADTYPE;
    switch (eco_size_of(t)) {
        case 1: { const uint8_t *a = p; uint8_t k = nd->v.u8; PAR_FOR for (size_t i = 0; i < n; i++) mask[i] = a[i] != k; break; }
        case 2: { const uint16_t *a = p; uint16_t k = nd->v.u16; PAR_FOR for (size_t i = 0; i < n; i++) mask[i] = a[i] != k; break; }
        case 4: { const uint32_t *a = p; uint32_t k = nd->v.u32; PAR_FOR for (size_t i = 0; i < n; i++) mask[i] = a[i] != k; break; }
        default: { const uint64_t *a = p; uint64_t k = nd->v.u64; PAR_FOR for (size_t i = 0; i < n; i++) mask[i] = a[i] != k; break; }
    }
    return ECO_OK;
}

int ecof_mask_select(int t, const void *p, const uint8_t *mask, size_t n, const eco_value *nd, void *out) {
    if (t < 0 || t >= ECO_NTYPES) return ECO_ERR_BADTYPE;
    if (!nd) { memcpy(out, p, n * eco_size_of(t)); return ECO_OK; }
    if (nd->ct != t) return ECO_ERR_BADTYPE;
    switch (eco_size_of(t)) {
        case 1: { const uint8_t *a = p; uint8_t *o = out; uint8_t k = nd->v.u8; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = mask[i] ? a[i] : k; break; }
        case 2: { const uint16_t *a = p; uint16_t *o = out; uint16_t k = nd->v.u16; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = mask[i] ? a[i] : k; break; }
        case 4: { const uint32_t *a = p; uint32_t *o = out; uint32_t k = nd->v.u32; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = mask[i] ? a[i] : k; break; }
        default: { const uint64_t *a = p; uint64_t *o = out; uint64_t k = nd->v.u64; PAR_FOR for (size_t i = 0; i < n; i++) o[i] = mask[i] ? a[i] : k; break; }
    }
    return ECO_OK;
}

/* ------------------------------------------------------------------ */
/* synthetic inputs: counter-based, reproducible on device. */
uint64_t eco_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

void eco_fill_u8(uint8_t *p, size_t n, uint64_t seed, uint64_t base, uint32_t lo, uint32_t hi) {
    uint64_t span = (uint64_t)hi - lo + 1;
    PAR_FOR for (size_t i = 0; i < n; i++) p[i] = (uint8_t)(lo + eco_splitmix64(seed ^ (base + i)) % span);
}

void eco_fill_u16(uint16_t *p, size_t n, uint64_t seed, uint64_t base, uint32_t lo, uint32_t hi) {
    uint64_t span = (uint64_t)hi - lo + 1;
    PAR_FOR for (size_t i = 0; i < n; i++) p[i] = (uint16_t)(lo + eco_splitmix64(seed ^ (base + i)) % span);
}
