// ec_abi.hip — the extern "C" compute surface of liberased_cells_hip.so (include/erased_cells.h):
// the type lattice and the type-erased dispatch of every kernel family except the four binary ops
// (ec_binop_*.hip) and the fused chains (ec_fused*.hip).  Devices, memory, streams: ec_runtime.hip.
//
// There is no CPU fallback anywhere in this library: without a HIP device every
// compute entry point returns EC_ERR_HIP / EC_ERR_NOT_INITIALIZED.
#include <hip/hip_runtime.h>


#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "ec_lattice.hpp"
#include "ec_map_kernels.hpp"
#include "ec_reduce_kernels.hpp"
#include "ec_runtime.hpp"

namespace ecd {

static_assert(kMaxReduceBlocks <= kFinalizeMaxParts, "the finalize kernels read at most kFinalizeMaxParts partials");

static ec_status ensure_init() { return ensure_ready(); }

// Where the last kernel of a synchronous-result call writes its 1-2 result words: the stream's pinned host words as the
// device sees them (zero-copy: the result is on the host when the stream has drained), or device scratch when the
// mapping is unavailable.  fetch_result waits for the stream (and copies first in the fallback case).
static int64_t* result_words(const Scratch& sc) { return sc.host_dev ? sc.host_dev : sc.dev_result(); }
static ec_status fetch_result(const Scratch& sc, int words, hipStream_t s) {
    if (!sc.host_dev) {
        ec_status st = check_hip(hipMemcpyAsync(sc.host, sc.dev_result(), words * sizeof(int64_t), hipMemcpyDeviceToHost, s), "hipMemcpyAsync");
        if (st != EC_OK) return st;
    }
    // (Polling the stream with hipStreamQuery before blocking was tried for the 5 µs kernels of fixture-sized rasters: 19.5 µs per ec_min_max
    // call against 15.1 µs — hipStreamSynchronize's own wait is the faster one; profiles/r04/sync_result_latency.txt.)
    return check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");
}

// Workgroups of `kernel` (BLOCK threads, no dynamic LDS) that fit on one CU at a time, at most `want`.
template <typename K>
static int resident_per_cu(K kernel, int block, int want) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        return want;
    }
    return nb < want ? nb : want;
}

// Workgroups a reduction launches at most: `per_cu` per CU (the launch shape's own default: as many as are resident
// at once, so the grid runs as one round) unless "reduce_bpc" overrides it; never more than the finalize kernels read.
static int reduce_cap(int per_cu) {
    const int knob = tuning().reduce_bpc;
    const long cap = long(device_cus()) * (knob > 0 ? knob : per_cu);
    return static_cast<int>(cap < kMaxReduceBlocks ? cap : kMaxReduceBlocks);
}

static inline hipStream_t S(ec_stream s) { return static_cast<hipStream_t>(s); }

template <typename Fn> struct pure_store : std::false_type {};
template <typename W> struct pure_store<FillFn<W>> : std::true_type {};

template <typename Fn, int U>
static void launch_map_u(const Fn& fn, size_t n, hipStream_t s) {
    const size_t groups = n / Fn::CPL;
    const size_t tiles = (groups + size_t(kBlock) * U - 1) / (size_t(kBlock) * U);
    const size_t stream_bytes[2] = {n * Fn::kIn0, n * Fn::kIn1};
    // ec_fill is a pure write stream with nothing to compute: unused dynamic LDS caps its resident workgroups per CU (the generators
    // load nothing either, but hash every cell — they need their occupancy: 576 -> 879 µs for 2^28 u8 cells under the same cap)
    const unsigned lds = (pure_store<Fn>::value ? static_cast<unsigned>(tuning().write_lds_kb.load())
                                                : (Fn::kIn0 != 0 ? static_cast<unsigned>(tuning().map_lds_kb.load()) : 0u)) << 10;
    k_map<Fn, U><<<grid_for(tiles), kBlock, lds, s>>>(fn, n, cache_plan(stream_bytes, 2));
}

template <typename Fn>
static ec_status launch_map(const Fn& fn, size_t n, bool aligned, hipStream_t s, const char* what) {
    if (n == 0) return EC_OK;
    if (!aligned) {
        k_map_cellwise<Fn><<<grid_capped((n + kBlock - 1) / kBlock, 8), kBlock, 0, s>>>(fn, n);
    } else {
        switch (tuning().map_u) {  // groups of 16 B per lane per tile
            case 1: launch_map_u<Fn, 1>(fn, n, s); break;
            case 4: launch_map_u<Fn, 4>(fn, n, s); break;
            default: launch_map_u<Fn, 2>(fn, n, s); break;
        }
    }
    return check_launch(what);
}

// ------------------------------------------------------------------ convert
template <typename Sx, typename Dx>
static ec_status launch_convert(const void* src, void* dst, size_t n, hipStream_t s) {
    if constexpr (ecl::can_fit_into(ecl::dtype_of<Sx>::value, ecl::dtype_of<Dx>::value) &&
                  ecl::dtype_of<Sx>::value != ecl::dtype_of<Dx>::value) {
        ConvertFn<Sx, Dx> fn{static_cast<const Sx*>(src), static_cast<Dx*>(dst)};
        return launch_map(fn, n, aligned16(src, dst, dst), s, "convert");
    } else {
        (void)src; (void)dst; (void)n; (void)s;
        return set_error(EC_ERR_ARG, "convert: pair not instantiated");
    }
}

static ec_status dispatch_convert(int st, const void* src, int dt, void* dst, size_t n, hipStream_t s) {
#define EC_ROW(SID, ST)                                                      \
    case SID:                                                                \
        switch (dt) {                                                        \
            case EC_U8: return launch_convert<ST, uint8_t>(src, dst, n, s);  \
            case EC_U16: return launch_convert<ST, uint16_t>(src, dst, n, s); \
            case EC_U32: return launch_convert<ST, uint32_t>(src, dst, n, s); \
            case EC_U64: return launch_convert<ST, uint64_t>(src, dst, n, s); \
            case EC_I8: return launch_convert<ST, int8_t>(src, dst, n, s);   \
            case EC_I16: return launch_convert<ST, int16_t>(src, dst, n, s); \
            case EC_I32: return launch_convert<ST, int32_t>(src, dst, n, s); \
            case EC_I64: return launch_convert<ST, int64_t>(src, dst, n, s); \
            case EC_F32: return launch_convert<ST, float>(src, dst, n, s);   \
            case EC_F64: return launch_convert<ST, double>(src, dst, n, s);  \
        }                                                                    \
        break;
    switch (st) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return set_error(EC_ERR_UNSUPPORTED_TYPE, "convert: bad dtype");
}

// ------------------------------------------------------------------ min/max
template <typename T>
static ec_status launch_min_max(const void* p, const uint8_t* mask, size_t n, int64_t* keys2_dev, hipStream_t s) {
    Scratch sc;
    ec_status st = get_scratch(s, &sc);
    if (st != EC_OK) return st;
    const T* tp = static_cast<const T*>(p);
    unsigned grid = 0;
    if (n > 0) {
        const bool al = aligned16(p, p, p) && (!mask || aligned_to(mask, 16 / sizeof(T)));
        const int cap = reduce_cap(8);  // the cell-wise kernel: 256-thread workgroups
        if (al) {
            const unsigned head = reduce_head(p, sizeof(T), n);
            const size_t groups = (n - head) / (16 / sizeof(T));
            // launch shape: 512 threads x 8 loads in flight by default; "reduce_shape" selects the A/B alternatives
            auto launch = [&](auto u_tag, auto block_tag, int per_cu) {
                constexpr int U = decltype(u_tag)::value, BLOCK = decltype(block_tag)::value;
                // as many workgroups per CU as are resident at once (the masked kernels of some types need more than
                // 64 VGPRs and fit 3, not 4, of these workgroups on a CU): the grid then runs as ONE round
                static const int resident[2] = {resident_per_cu(k_min_max_partials<T, false, U, BLOCK>, BLOCK, per_cu),
                                                resident_per_cu(k_min_max_partials<T, true, U, BLOCK>, BLOCK, per_cu)};
                const int cap2 = reduce_cap(resident[mask ? 1 : 0]);
                size_t tiles = (groups + size_t(BLOCK) * U - 1) / (size_t(BLOCK) * U);
                if (tiles < 1) tiles = 1;
                grid = static_cast<unsigned>(tiles < size_t(cap2) ? tiles : size_t(cap2));
                int64_t* direct = grid == 1 ? keys2_dev : nullptr;  // one workgroup: it writes the result itself
                const size_t stream_bytes[2] = {n * sizeof(T), mask ? n : 0};
                const unsigned hp = head | (cache_plan(stream_bytes, 2) << 8);  // leading cells + load policy
                if (mask) k_min_max_partials<T, true, U, BLOCK><<<grid, BLOCK, 0, s>>>(tp, mask, n, sc.dev, hp, direct);
                else k_min_max_partials<T, false, U, BLOCK><<<grid, BLOCK, 0, s>>>(tp, nullptr, n, sc.dev, hp, direct);
                return direct != nullptr;
            };
            using std::integral_constant;
            bool direct = false;
            switch (tuning().reduce_shape) {
                case 1: direct = launch(integral_constant<int, 16>{}, integral_constant<int, 512>{}, 4); break;
                case 2: direct = launch(integral_constant<int, 8>{}, integral_constant<int, 256>{}, 8); break;
                case 3: direct = launch(integral_constant<int, 8>{}, integral_constant<int, 1024>{}, 2); break;
                case 4: direct = launch(integral_constant<int, 4>{}, integral_constant<int, 512>{}, 4); break;
                default: direct = launch(integral_constant<int, kReduceU>{}, integral_constant<int, kRBlock>{}, 4); break;
            }
            if (direct) return check_launch("min_max(single workgroup)");
        } else {
            size_t blocks = (n + kBlock - 1) / kBlock;
            grid = static_cast<unsigned>(blocks < size_t(cap) ? blocks : size_t(cap));
            if (mask) k_min_max_partials_cellwise<T, true><<<grid, kBlock, 0, s>>>(tp, mask, n, sc.dev);
            else k_min_max_partials_cellwise<T, false><<<grid, kBlock, 0, s>>>(tp, nullptr, n, sc.dev);
        }
        st = check_launch("min_max(partials)");
        if (st != EC_OK) return st;
    }
    // sentinels (T::MAX, T::MIN): src/buffer.rs:170, finite for floats (src/ctype.rs:158-179)
    k_min_max_finalize<<<1, kFinalizeBlock, 0, s>>>(sc.dev, static_cast<int>(grid), order_key<T>(Limits<T>::hi),
                                            order_key<T>(Limits<T>::lo), keys2_dev);
    return check_launch("min_max(finalize)");
}

static ec_status dispatch_min_max(int t, const void* p, const uint8_t* mask, size_t n, int64_t* keys2_dev, hipStream_t s) {
#define EC_ROW(ID, T) case ID: return launch_min_max<T>(p, mask, n, keys2_dev, s);
    switch (t) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return set_error(EC_ERR_UNSUPPORTED_TYPE, "min_max: bad dtype");
}

template <typename T>
static void decode_keys(const int64_t keys2[2], ec_value* mn, ec_value* mx) {
    T a = key_value<T>(~keys2[0]), b = key_value<T>(keys2[1]);
    std::memset(mn, 0, sizeof *mn);
    std::memset(mx, 0, sizeof *mx);
    mn->dtype = mx->dtype = static_cast<uint8_t>(ecl::dtype_of<T>::value);
    std::memcpy(&mn->v, &a, sizeof a);
    std::memcpy(&mx->v, &b, sizeof b);
}

}  // namespace ecd

using namespace ecd;

// =================================================================== lattice
extern "C" ec_dtype ec_union(ec_dtype a, ec_dtype b) {
    return (ecl::valid(a) && ecl::valid(b)) ? static_cast<ec_dtype>(ecl::union_of(a, b)) : static_cast<ec_dtype>(EC_F64);
}
extern "C" int32_t ec_can_fit_into(ec_dtype src, ec_dtype dst) {
    return ecl::valid(src) && ecl::valid(dst) && ecl::can_fit_into(src, dst);
}
extern "C" size_t ec_size_of(ec_dtype t) { return ecl::size_of(t); }
extern "C" ec_dtype ec_neg_result_type(ec_dtype t) { return static_cast<ec_dtype>(ecl::neg_result(t)); }

template <typename T>
static void put_value(ec_value* out, int dtype, T x) {
    std::memset(out, 0, sizeof *out);
    out->dtype = static_cast<uint8_t>(dtype);
    std::memcpy(&out->v, &x, sizeof x);
}

extern "C" ec_status ec_min_value(ec_dtype t, ec_value* out) {
    if (!out) return set_error(EC_ERR_ARG, "ec_min_value: null out");
#define EC_ROW(ID, T) case ID: put_value<T>(out, ID, Limits<T>::lo); return EC_OK;
    switch (t) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_min_value: bad dtype %d", int(t));
}
extern "C" ec_status ec_max_value(ec_dtype t, ec_value* out) {
    if (!out) return set_error(EC_ERR_ARG, "ec_max_value: null out");
#define EC_ROW(ID, T) case ID: put_value<T>(out, ID, Limits<T>::hi); return EC_OK;
    switch (t) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_max_value: bad dtype %d", int(t));
}
extern "C" ec_status ec_nodata_default(ec_dtype t, ec_value* out) {
    if (!out) return set_error(EC_ERR_ARG, "ec_nodata_default: null out");
    if (t == EC_F32) { put_value<uint32_t>(out, t, 0x7fc00000u); return EC_OK; }           // f32::NAN
    if (t == EC_F64) { put_value<uint64_t>(out, t, 0x7ff8000000000000ull); return EC_OK; }  // f64::NAN
    return ec_min_value(t, out);                                                            // <int>::MIN
}

template <typename T>
static double value_as_f64(const ec_value* v) { T x; std::memcpy(&x, &v->v, sizeof x); return static_cast<double>(x); }

extern "C" double ec_value_to_f64(const ec_value* v) {
    if (!v) return 0.0;
#define EC_ROW(ID, T) case ID: return value_as_f64<T>(v);
    switch (v->dtype) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return 0.0;
}

template <typename Sx, typename Dx>
static void value_cast(const ec_value* v, ec_value* out) {
    Sx x; std::memcpy(&x, &v->v, sizeof x);
    put_value<Dx>(out, ecl::dtype_of<Dx>::value, static_cast<Dx>(x));
}

extern "C" ec_status ec_value_convert(const ec_value* v, ec_dtype dst, ec_value* out) {
    if (!v || !out) return set_error(EC_ERR_ARG, "ec_value_convert: null argument");
    if (!ecl::valid(v->dtype) || !ecl::valid(dst)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_value_convert: bad dtype");
    if (!ecl::can_fit_into(v->dtype, dst)) return set_narrowing(v->dtype, dst);  // value.rs:79-81
    if (v->dtype == dst) { *out = *v; return EC_OK; }                            // value.rs:83-85
#define EC_ROW(SID, ST)                                                \
    case SID:                                                          \
        switch (dst) {                                                 \
            case EC_U8: value_cast<ST, uint8_t>(v, out); return EC_OK; \
            case EC_U16: value_cast<ST, uint16_t>(v, out); return EC_OK; \
            case EC_U32: value_cast<ST, uint32_t>(v, out); return EC_OK; \
            case EC_U64: value_cast<ST, uint64_t>(v, out); return EC_OK; \
            case EC_I8: value_cast<ST, int8_t>(v, out); return EC_OK;  \
            case EC_I16: value_cast<ST, int16_t>(v, out); return EC_OK; \
            case EC_I32: value_cast<ST, int32_t>(v, out); return EC_OK; \
            case EC_I64: value_cast<ST, int64_t>(v, out); return EC_OK; \
            case EC_F32: value_cast<ST, float>(v, out); return EC_OK;  \
            case EC_F64: value_cast<ST, double>(v, out); return EC_OK; \
        }                                                              \
        break;
    switch (v->dtype) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_value_convert: bad dtype");
}

extern "C" ec_status ec_shard_range(uint64_t n_rows, uint64_t n_cols, uint32_t shard, uint32_t n_shards,
                                    uint64_t* cell_offset, uint64_t* cell_len) {
    if (!cell_offset || !cell_len || n_shards == 0 || shard >= n_shards)
        return set_error(EC_ERR_ARG, "ec_shard_range: shard %u of %u", shard, n_shards);
    const uint64_t q = n_rows / n_shards, rem = n_rows % n_shards;
    const uint64_t row0 = uint64_t(shard) * q + (shard < rem ? shard : rem);
    const uint64_t rows = q + (shard < rem ? 1 : 0);
    *cell_offset = row0 * n_cols;
    *cell_len = rows * n_cols;
    return EC_OK;
}

// =================================================================== arithmetic
#define EC_REQUIRE_INIT()               \
    do {                                \
        ec_status st_ = ensure_init();  \
        if (st_ != EC_OK) return st_;   \
    } while (0)

extern "C" ec_status ec_binop(ec_op op, ec_dtype lt, const void* l, ec_dtype rt, const void* r, size_t n, double* out,
                              ec_stream stream) {
    EC_REQUIRE_INIT();
    if (n == 0) return EC_OK;
    if (!l || !r || !out) return set_error(EC_ERR_ARG, "ec_binop: null pointer");
    switch (op) {
        case EC_ADD: return dispatch_binop<EC_ADD>(lt, l, rt, r, n, out, S(stream));
        case EC_SUB: return dispatch_binop<EC_SUB>(lt, l, rt, r, n, out, S(stream));
        case EC_MUL: return dispatch_binop<EC_MUL>(lt, l, rt, r, n, out, S(stream));
        case EC_DIV: return dispatch_binop<EC_DIV>(lt, l, rt, r, n, out, S(stream));
    }
    return set_error(EC_ERR_ARG, "ec_binop: bad op %d", int(op));
}

extern "C" ec_status ec_binop_scalar(ec_op op, ec_dtype lt, const void* l, size_t n, const ec_value* rhs, double* out,
                                     ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!rhs || !ecl::valid(rhs->dtype)) return set_error(EC_ERR_ARG, "ec_binop_scalar: bad rhs");
    if (n == 0) return EC_OK;
    if (!l || !out) return set_error(EC_ERR_ARG, "ec_binop_scalar: null pointer");
    // unify + to_f64 of the scalar (value.rs:206-207) happens once, here
    const double s = ec_value_to_f64(rhs);
    switch (op) {
        case EC_ADD: return dispatch_binop_scalar<EC_ADD>(lt, l, s, n, out, S(stream));
        case EC_SUB: return dispatch_binop_scalar<EC_SUB>(lt, l, s, n, out, S(stream));
        case EC_MUL: return dispatch_binop_scalar<EC_MUL>(lt, l, s, n, out, S(stream));
        case EC_DIV: return dispatch_binop_scalar<EC_DIV>(lt, l, s, n, out, S(stream));
    }
    return set_error(EC_ERR_ARG, "ec_binop_scalar: bad op %d", int(op));
}

extern "C" ec_status ec_masked_binop(ec_op op, ec_dtype lt, const void* l, const uint8_t* lmask, ec_dtype rt, const void* r,
                                     const uint8_t* rmask, size_t n, double* out, uint8_t* out_mask, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (n == 0) return EC_OK;
    if (!l || !r || !out || !lmask || !rmask || !out_mask) return set_error(EC_ERR_ARG, "ec_masked_binop: null pointer");
    switch (op) {
        case EC_ADD: return dispatch_masked_binop<EC_ADD>(lt, l, lmask, rt, r, rmask, n, out, out_mask, S(stream));
        case EC_SUB: return dispatch_masked_binop<EC_SUB>(lt, l, lmask, rt, r, rmask, n, out, out_mask, S(stream));
        case EC_MUL: return dispatch_masked_binop<EC_MUL>(lt, l, lmask, rt, r, rmask, n, out, out_mask, S(stream));
        case EC_DIV: return dispatch_masked_binop<EC_DIV>(lt, l, lmask, rt, r, rmask, n, out, out_mask, S(stream));
    }
    return set_error(EC_ERR_ARG, "ec_masked_binop: bad op %d", int(op));
}

extern "C" ec_status ec_neg(ec_dtype t, const void* in, size_t n, void* out, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!ecl::valid(t)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_neg: bad dtype %d", int(t));
    if (n == 0) return EC_OK;
    if (!in || !out) return set_error(EC_ERR_ARG, "ec_neg: null pointer");
#define EC_ROW(ID, T)                                                                             \
    case ID: {                                                                                    \
        NegFn<T> fn{static_cast<const T*>(in), static_cast<typename NegOut<T>::type*>(out)};      \
        return launch_map(fn, n, aligned16(in, out, out), S(stream), "neg");                      \
    }
    switch (t) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return EC_OK;
}

extern "C" ec_status ec_convert(ec_dtype st, const void* src, ec_dtype dt, void* dst, size_t n, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!ecl::valid(st) || !ecl::valid(dt)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_convert: bad dtype");
    if (!ecl::can_fit_into(st, dt)) return set_narrowing(st, dt);  // buffer.rs:157-159: before touching data
    if (n == 0) return EC_OK;
    if (!src || !dst) return set_error(EC_ERR_ARG, "ec_convert: null pointer");
    if (st == dt) return ec_copy(dst, src, n * ecl::size_of(st), stream);  // buffer.rs:151-153
    return dispatch_convert(st, src, dt, dst, n, S(stream));
}

template <typename W>
static ec_status fill_w(void* dst, size_t n, const ec_value* v, hipStream_t s) {
    W w; std::memcpy(&w, &v->v, sizeof w);
    FillFn<W> fn{static_cast<W*>(dst), w};
    return launch_map(fn, n, aligned16(dst, dst, dst), s, "fill");
}

extern "C" ec_status ec_fill(ec_dtype t, void* dst, size_t n, const ec_value* value, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!ecl::valid(t) || !value || value->dtype != t) return set_error(EC_ERR_ARG, "ec_fill: value dtype must equal t");
    if (n == 0) return EC_OK;
    if (!dst) return set_error(EC_ERR_ARG, "ec_fill: null pointer");
    switch (ecl::size_of(t)) {
        case 1: return fill_w<uint8_t>(dst, n, value, S(stream));
        case 2: return fill_w<uint16_t>(dst, n, value, S(stream));
        case 4: return fill_w<uint32_t>(dst, n, value, S(stream));
        default: return fill_w<uint64_t>(dst, n, value, S(stream));
    }
}

// =================================================================== min/max
extern "C" ec_status ec_min_max_keys(ec_dtype t, const void* p, const uint8_t* mask_or_null, size_t n, int64_t* keys2_dev,
                                     ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!keys2_dev || (n > 0 && !p)) return set_error(EC_ERR_ARG, "ec_min_max_keys: null pointer");
    return dispatch_min_max(t, p, mask_or_null, n, keys2_dev, S(stream));
}

extern "C" ec_status ec_min_max_decode(ec_dtype t, const int64_t keys2_host[2], ec_value* mn, ec_value* mx) {
    if (!keys2_host || !mn || !mx) return set_error(EC_ERR_ARG, "ec_min_max_decode: null pointer");
#define EC_ROW(ID, T) case ID: decode_keys<T>(keys2_host, mn, mx); return EC_OK;
    switch (t) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_min_max_decode: bad dtype %d", int(t));
}

extern "C" ec_status ec_min_max(ec_dtype t, const void* p, const uint8_t* mask_or_null, size_t n, ec_value* mn, ec_value* mx,
                                ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!mn || !mx || (n > 0 && !p)) return set_error(EC_ERR_ARG, "ec_min_max: null pointer");
    Scratch sc;
    ec_status st = get_scratch(S(stream), &sc);
    if (st != EC_OK) return st;
    std::lock_guard<std::mutex> turn(*sc.mu);  // the pinned result words are per stream: host threads sharing it take turns
    st = dispatch_min_max(t, p, mask_or_null, n, result_words(sc), S(stream));  // the last kernel writes the pinned words
    if (st != EC_OK) return st;
    st = fetch_result(sc, 2, S(stream));
    if (st != EC_OK) return st;
    return ec_min_max_decode(t, sc.host, mn, mx);
}

// =================================================================== Ord / Eq
template <typename W>
static ec_status first_diff_w(const void* l, const void* r, size_t n, const Scratch& sc, hipStream_t s, unsigned* grid_out) {
    const int cap = reduce_cap(4);
    const bool al = aligned16(l, r, r);
    const unsigned head = al ? reduce_head(l, sizeof(W), n) : 0u;
    size_t tiles = al ? ((n - head) / (16 / sizeof(W)) + size_t(kRBlock) * kReduceU - 1) / (size_t(kRBlock) * kReduceU) : (n + kRBlock - 1) / kRBlock;
    if (tiles < 1) tiles = 1;
    const unsigned grid = static_cast<unsigned>(tiles < size_t(cap) ? tiles : size_t(cap));
    const size_t stream_bytes[2] = {n * sizeof(W), n * sizeof(W)};
    k_first_diff_partials<W, kReduceU><<<grid, kRBlock, 0, s>>>(static_cast<const W*>(l), static_cast<const W*>(r), n,
                                                               reinterpret_cast<uint64_t*>(sc.dev), al, head | (cache_plan(stream_bytes, 2) << 8));
    *grid_out = grid;
    return check_launch("first_diff(partials)");
}

extern "C" ec_status ec_first_difference(ec_dtype t, const void* l, const void* r, size_t n, uint64_t* index, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!ecl::valid(t)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_first_difference: bad dtype %d", int(t));
    if (!index || (n > 0 && (!l || !r))) return set_error(EC_ERR_ARG, "ec_first_difference: null pointer");
    *index = n;
    if (n == 0) return EC_OK;
    Scratch sc;
    ec_status st = get_scratch(S(stream), &sc);
    if (st != EC_OK) return st;
    std::lock_guard<std::mutex> turn(*sc.mu);
    unsigned grid = 0;
    switch (ecl::size_of(t)) {
        case 1: st = first_diff_w<uint8_t>(l, r, n, sc, S(stream), &grid); break;
        case 2: st = first_diff_w<uint16_t>(l, r, n, sc, S(stream), &grid); break;
        case 4: st = first_diff_w<uint32_t>(l, r, n, sc, S(stream), &grid); break;
        default: st = first_diff_w<uint64_t>(l, r, n, sc, S(stream), &grid); break;
    }
    if (st != EC_OK) return st;
    k_first_diff_finalize<<<1, kFinalizeBlock, 0, S(stream)>>>(reinterpret_cast<const uint64_t*>(sc.dev), static_cast<int>(grid),
                                                       reinterpret_cast<uint64_t*>(result_words(sc)));
    st = check_launch("first_diff(finalize)");
    if (st != EC_OK) return st;
    st = fetch_result(sc, 1, S(stream));
    if (st != EC_OK) return st;
    const uint64_t first = static_cast<uint64_t>(sc.host[0]);
    *index = first == ~0ull ? n : first;
    return EC_OK;
}

template <typename T>
static int order_cmp(const void* a, const void* b) {
    T x, y;
    std::memcpy(&x, a, sizeof x);
    std::memcpy(&y, b, sizeof y);
    const int64_t kx = order_key<T>(x), ky = order_key<T>(y);
    return (kx > ky) - (kx < ky);
}

extern "C" ec_status ec_buffer_cmp(ec_dtype lt, const void* l, size_t nl, ec_dtype rt, const void* r, size_t nr,
                                   int32_t* ordering, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!ordering) return set_error(EC_ERR_ARG, "ec_buffer_cmp: null result pointer");
    if (!ecl::valid(lt) || !ecl::valid(rt)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_buffer_cmp: bad dtype");
    if (lt != rt) { *ordering = lt < rt ? -1 : 1; return EC_OK; }  // cell type first (buffer.rs:391-398)
    const size_t n = nl < nr ? nl : nr;
    uint64_t idx = n;
    ec_status st = ec_first_difference(lt, l, r, n, &idx, stream);
    if (st != EC_OK) return st;
    if (idx >= n) { *ordering = (nl > nr) - (nl < nr); return EC_OK; }  // common prefix equal: length decides (:416)
    unsigned char a[8], b[8];
    const size_t sz = ecl::size_of(lt);
    st = ec_download(a, static_cast<const char*>(l) + idx * sz, sz, stream);
    if (st != EC_OK) return st;
    st = ec_download(b, static_cast<const char*>(r) + idx * sz, sz, stream);
    if (st != EC_OK) return st;
#define EC_ROW(ID, T) case ID: *ordering = order_cmp<T>(a, b); return EC_OK;
    switch (lt) { EC_WITH_CT(EC_ROW) }
#undef EC_ROW
    return EC_OK;
}

// =================================================================== masks
template <typename W>
static ec_status mask_from_nd_w(const void* p, size_t n, const ec_value* nd, uint8_t* mask, hipStream_t s) {
    W w; std::memcpy(&w, &nd->v, sizeof w);
    MaskFromNodataFn<W> fn{static_cast<const W*>(p), mask, w};
    const bool al = aligned16(p, p, p) && aligned_to(mask, 16 / sizeof(W));
    return launch_map(fn, n, al, s, "mask_from_nodata");
}

extern "C" ec_status ec_mask_from_nodata(ec_dtype t, const void* p, size_t n, const ec_value* nd_or_null, uint8_t* mask,
                                         ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!ecl::valid(t)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_mask_from_nodata: bad dtype %d", int(t));
    if (nd_or_null && nd_or_null->dtype != t) return set_error(EC_ERR_ARG, "ec_mask_from_nodata: nodata dtype must equal t");
    if (n == 0) return EC_OK;
    if (!p || !mask) return set_error(EC_ERR_ARG, "ec_mask_from_nodata: null pointer");
    if (!nd_or_null)  // NoData::None: nothing matches (nodata.rs:43-48) -> all true
        return check_hip(hipMemsetAsync(mask, 1, n, S(stream)), "hipMemsetAsync");
    switch (ecl::size_of(t)) {
        case 1: return mask_from_nd_w<uint8_t>(p, n, nd_or_null, mask, S(stream));
        case 2: return mask_from_nd_w<uint16_t>(p, n, nd_or_null, mask, S(stream));
        case 4: return mask_from_nd_w<uint32_t>(p, n, nd_or_null, mask, S(stream));
        default: return mask_from_nd_w<uint64_t>(p, n, nd_or_null, mask, S(stream));
    }
}

template <typename W>
static ec_status mask_select_w(const void* p, const uint8_t* mask, size_t n, const ec_value* nd, void* out, hipStream_t s) {
    W w; std::memcpy(&w, &nd->v, sizeof w);
    MaskSelectFn<W> fn{static_cast<const W*>(p), mask, static_cast<W*>(out), w};
    const bool al = aligned16(p, out, out) && aligned_to(mask, 16 / sizeof(W));
    return launch_map(fn, n, al, s, "mask_select");
}

extern "C" ec_status ec_mask_select(ec_dtype t, const void* p, const uint8_t* mask, size_t n, const ec_value* nd_or_null,
                                    void* out, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!ecl::valid(t)) return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_mask_select: bad dtype %d", int(t));
    if (nd_or_null && nd_or_null->dtype != t) return set_error(EC_ERR_ARG, "ec_mask_select: nodata dtype must equal t");
    if (n == 0) return EC_OK;
    if (!p || !out) return set_error(EC_ERR_ARG, "ec_mask_select: null pointer");
    if (!nd_or_null) return ec_copy(out, p, n * ecl::size_of(t), stream);  // masked_buffer.rs:149-151
    if (!mask) return set_error(EC_ERR_ARG, "ec_mask_select: null mask");
    switch (ecl::size_of(t)) {
        case 1: return mask_select_w<uint8_t>(p, mask, n, nd_or_null, out, S(stream));
        case 2: return mask_select_w<uint16_t>(p, mask, n, nd_or_null, out, S(stream));
        case 4: return mask_select_w<uint32_t>(p, mask, n, nd_or_null, out, S(stream));
        default: return mask_select_w<uint64_t>(p, mask, n, nd_or_null, out, S(stream));
    }
}

extern "C" ec_status ec_mask_and(const uint8_t* l, const uint8_t* r, size_t n, uint8_t* out, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (n == 0) return EC_OK;
    if (!l || !r || !out) return set_error(EC_ERR_ARG, "ec_mask_and: null pointer");
    MaskBin<0> fn{l, r, out};
    return launch_map(fn, n, aligned16(l, r, out), S(stream), "mask_and");
}
extern "C" ec_status ec_mask_or(const uint8_t* l, const uint8_t* r, size_t n, uint8_t* out, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (n == 0) return EC_OK;
    if (!l || !r || !out) return set_error(EC_ERR_ARG, "ec_mask_or: null pointer");
    MaskBin<1> fn{l, r, out};
    return launch_map(fn, n, aligned16(l, r, out), S(stream), "mask_or");
}
extern "C" ec_status ec_mask_not(const uint8_t* m, size_t n, uint8_t* out, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (n == 0) return EC_OK;
    if (!m || !out) return set_error(EC_ERR_ARG, "ec_mask_not: null pointer");
    MaskNot fn{m, out};
    return launch_map(fn, n, aligned16(m, out, out), S(stream), "mask_not");
}

extern "C" ec_status ec_mask_counts_device(const uint8_t* m, size_t n, uint64_t* counts2_dev, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!counts2_dev || (n > 0 && !m)) return set_error(EC_ERR_ARG, "ec_mask_counts_device: null pointer");
    Scratch sc;
    ec_status st = get_scratch(S(stream), &sc);
    if (st != EC_OK) return st;
    unsigned grid = 0;
    if (n > 0) {
        const int cap = reduce_cap(4);
        const bool al = aligned_to(m, 16);
        const unsigned head = al ? reduce_head(m, 1, n) : 0u;
        size_t tiles = al ? ((n - head) / 16 + size_t(kRBlock) * kReduceU - 1) / (size_t(kRBlock) * kReduceU) : (n + kRBlock - 1) / kRBlock;
        if (tiles < 1) tiles = 1;
        grid = static_cast<unsigned>(tiles < size_t(cap) ? tiles : size_t(cap));
        uint64_t* direct = grid == 1 ? counts2_dev : nullptr;  // one workgroup: it writes the result itself
        // more workgroups: the last one to add its count to the stream's accumulator word writes the result (one launch) — for masks below
        // 2^29 cells.  Measured, rotating masks, every byte from HBM (profiles/r04/mask_counts_one_launch.md): 4096² 9.5 -> 7.5 µs, 16384²
        // 49.5 -> 47.7 µs, but 32768² 158.4 -> 166.7 µs: the 1,024 returning atomics on one address queue behind the channel's reads when
        // every workgroup is still streaming a gigabyte.  Big masks keep the finalize launch, which they do not feel (2 %); knob
        // `counts_one_launch`: 0 never, 1 (default) below 2^29 cells, 2 always (below 2^40: ticket and sum share a 64-bit word).
        const int one = tuning().counts_one_launch.load();
        uint64_t* acc = (!direct && ((one == 1 && n < (size_t(1) << 29)) || (one >= 2 && n < (size_t(1) << 40)))) ? reinterpret_cast<uint64_t*>(sc.dev_acc()) : nullptr;
        const size_t stream_bytes[1] = {n};
        k_mask_count_partials<kReduceU><<<grid, kRBlock, 0, S(stream)>>>(m, n, reinterpret_cast<uint64_t*>(sc.dev), al,
                                                                        head | (cache_plan(stream_bytes, 1) << 8), direct, acc, counts2_dev);
        st = check_launch("mask_counts(partials)");
        if (st != EC_OK || direct || acc) return st;
    }
    k_mask_count_finalize<<<1, kFinalizeBlock, 0, S(stream)>>>(reinterpret_cast<const uint64_t*>(sc.dev), static_cast<int>(grid), n, counts2_dev);
    return check_launch("mask_counts(finalize)");
}

extern "C" ec_status ec_mask_counts(const uint8_t* m, size_t n, uint64_t* n_true, uint64_t* n_false, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (!n_true || !n_false) return set_error(EC_ERR_ARG, "ec_mask_counts: null pointer");
    Scratch sc;
    ec_status st = get_scratch(S(stream), &sc);
    if (st != EC_OK) return st;
    std::lock_guard<std::mutex> turn(*sc.mu);
    st = ec_mask_counts_device(m, n, reinterpret_cast<uint64_t*>(result_words(sc)), stream);
    if (st != EC_OK) return st;
    st = fetch_result(sc, 2, S(stream));
    if (st != EC_OK) return st;
    *n_true = static_cast<uint64_t>(sc.host[0]);
    *n_false = static_cast<uint64_t>(sc.host[1]);
    return EC_OK;
}

// =================================================================== synthetic inputs
template <typename T>
static ec_status synth_t(void* dst, size_t n, uint64_t seed, uint64_t base, double lo, double hi, hipStream_t s) {
    SynthFn<T> fn{static_cast<T*>(dst), seed, base, 1, lo, hi - lo};
    if constexpr (!is_fp<T>::value) fn.span = static_cast<uint64_t>(hi) - static_cast<uint64_t>(lo) + 1;
    return launch_map(fn, n, aligned16(dst, dst, dst), s, "synth_fill");
}

extern "C" ec_status ec_synth_fill(ec_dtype t, void* dst, size_t n, uint64_t seed, uint64_t base, double lo, double hi,
                                   ec_stream stream) {
    EC_REQUIRE_INIT();
    if (n == 0) return EC_OK;
    if (!dst) return set_error(EC_ERR_ARG, "ec_synth_fill: null pointer");
    switch (t) {
        case EC_U8: return synth_t<uint8_t>(dst, n, seed, base, lo, hi, S(stream));
        case EC_U16: return synth_t<uint16_t>(dst, n, seed, base, lo, hi, S(stream));
        case EC_U32: return synth_t<uint32_t>(dst, n, seed, base, lo, hi, S(stream));
        case EC_F32: return synth_t<float>(dst, n, seed, base, lo, hi, S(stream));
        case EC_F64: return synth_t<double>(dst, n, seed, base, lo, hi, S(stream));
    }
    return set_error(EC_ERR_UNSUPPORTED_TYPE, "ec_synth_fill: dtype %d not supported", int(t));
}

extern "C" ec_status ec_synth_mask(uint8_t* dst, size_t n, uint64_t seed, uint64_t base, uint32_t pct_nodata, ec_stream stream) {
    EC_REQUIRE_INIT();
    if (n == 0) return EC_OK;
    if (!dst) return set_error(EC_ERR_ARG, "ec_synth_mask: null pointer");
    SynthMaskFn fn{dst, seed, base, pct_nodata};
    return launch_map(fn, n, aligned16(dst, dst, dst), S(stream), "synth_mask");
}
