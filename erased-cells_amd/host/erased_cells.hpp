// erased_cells.hpp — C++ host mirror of the `erased-cells` API over the C ABI.
//
// The reference is a Rust crate; this image has no Rust toolchain, so the
// compiled host side above include/erased_cells.h is this header (C++17,
// header-only, links only liberased_cells_hip.so).  It keeps the reference's
// names, argument meaning and error behaviour:
//
//   CellType / CellEncoding / with_ct!   src/ctype.rs, src/encoding.rs, src/lib.rs:85-101
//   CellValue                            src/value.rs
//   CellBuffer + BufferOps + std::ops    src/buffer.rs, src/lib.rs:104-163
//   Mask / MaskedCellBuffer / NoData<T>  src/masked/*.rs
//   Error::NarrowingError                src/error.rs:14-15  (thrown as NarrowingError)
//
// What stays on the host — exactly what the reference's host code decides once
// per operation: the dtype-erased dispatch tag, zip truncation (buffer.rs:327),
// "empty result is UInt8" (buffer.rs:233-234), length asserts (panics become
// std::logic_error).  Every per-cell loop body is one call into the HIP library;
// buffers stay resident in HBM between operations (from_vec uploads once,
// to_vec downloads).  Scalar CellValue arithmetic is host arithmetic in the
// reference too (one f64 op) and stays so here.
//
// Rust ownership maps to C++ as: `&CellBuffer` -> const CellBuffer&, owned ->
// by value / rvalue; Clone is explicit (`clone()`), copies are deleted.
#pragma once

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "erased_cells.h"

namespace erased_cells {

// ---------------------------------------------------------------- CellType (src/ctype.rs:11-20)
enum class CellType : uint8_t { UInt8, UInt16, UInt32, UInt64, Int8, Int16, Int32, Int64, Float32, Float64 };

// with_ct! (src/lib.rs:85-101): X(Variant, primitive)
#define EC_HOST_WITH_CT(X) \
    X(UInt8, uint8_t) X(UInt16, uint16_t) X(UInt32, uint32_t) X(UInt64, uint64_t) X(Int8, int8_t) \
    X(Int16, int16_t) X(Int32, int32_t) X(Int64, int64_t) X(Float32, float) X(Float64, double)

inline const char* to_string(CellType ct) {
    static const char* n[] = {"UInt8", "UInt16", "UInt32", "UInt64", "Int8", "Int16", "Int32", "Int64", "Float32", "Float64"};
    return n[static_cast<int>(ct)];
}

// ---------------------------------------------------------------- Debug rendering (host only)
// `format!("{:?}", x)` of a Rust primitive: integers plain; floats as the shortest digits that round-trip for
// their width, decimal with at least one fractional digit for 1e-4 <= |x| < 1e16 (and zero), scientific
// (`1e16`, `1.5e-7`) outside, `NaN` / `inf` / `-inf`; bools `true` / `false`.
template <typename T> inline std::string rust_debug(T x) {
    if constexpr (std::is_same<T, bool>::value) {
        return x ? "true" : "false";
    } else if constexpr (std::is_integral<T>::value) {
        return std::to_string(+x);
    } else {
        if (x != x) return "NaN";
        if (std::isinf(x)) return x > 0 ? "inf" : "-inf";
        char buf[64];
        const T ax = x < 0 ? -x : x;
        if (ax == 0 || (ax >= T(1e-4) && ax < T(1e16))) {
            auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::fixed);  // shortest round-trip digits
            std::string out(buf, r.ptr);
            if (out.find('.') == std::string::npos) out += ".0";
            return out;
        }
        auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);
        std::string sci(buf, r.ptr);  // d[.ddd]e[+-]XX
        const size_t e = sci.find('e');
        return sci.substr(0, e) + "e" + std::to_string(std::stoi(sci.substr(e + 1)));
    }
}
// `Elided` (src/lib.rs:165-192): more than 10 items render as the first five, `, ... `, the last five.
inline std::string elided(const std::vector<std::string>& items) {
    std::string out;
    auto join = [&](size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) { if (i > a) out += ", "; out += items[i]; }
    };
    if (items.size() > 10) { join(0, 5); out += ", ... "; join(items.size() - 5, items.size()); }
    else join(0, items.size());
    return out;
}
inline std::vector<CellType> cell_types() {  // CellType::iter (ctype.rs:47-52)
    std::vector<CellType> v;
    for (int i = 0; i < EC_NTYPES; ++i) v.push_back(static_cast<CellType>(i));
    return v;
}

// ---------------------------------------------------------------- errors (src/error.rs)
struct Error : std::runtime_error {
    ec_status status;
    Error(ec_status s, const std::string& m) : std::runtime_error(m), status(s) {}
};
struct NarrowingError : Error {  // Error::NarrowingError{src,dst}
    CellType src, dst;
    NarrowingError(CellType s, CellType d)
        : Error(EC_ERR_NARROWING, std::string("Invalid narrowing from cell-type ") + to_string(s) + " to " + to_string(d)),
          src(s), dst(d) {}
};
struct ParseError : Error {  // Error::ParseError (ctype.rs:37)
    explicit ParseError(const std::string& s) : Error(EC_ERR_ARG, "Unable to parse " + s + " as a CellType") {}
};

inline void check(ec_status st) {
    if (st == EC_OK) return;
    if (st == EC_ERR_NARROWING) {
        ec_dtype s = 0, d = 0;
        ec_last_narrowing(&s, &d);
        throw NarrowingError(static_cast<CellType>(s), static_cast<CellType>(d));
    }
    throw Error(st, ec_last_error_string());
}

inline CellType cell_type_from_str(const std::string& s) {  // FromStr (ctype.rs:29-43)
    for (CellType ct : cell_types())
        if (s == to_string(ct)) return ct;
    throw ParseError(s);
}
inline bool is_integral(CellType ct) { return ct != CellType::Float32 && ct != CellType::Float64; }
inline bool is_signed(CellType ct) { return static_cast<int>(ct) >= EC_I8; }
inline size_t size_of(CellType ct) { return ec_size_of(static_cast<ec_dtype>(ct)); }
inline CellType union_of(CellType a, CellType b) {  // CellType::union (ctype.rs:99-126)
    return static_cast<CellType>(ec_union(static_cast<ec_dtype>(a), static_cast<ec_dtype>(b)));
}
inline bool can_fit_into(CellType a, CellType b) {  // ctype.rs:129-131
    return ec_can_fit_into(static_cast<ec_dtype>(a), static_cast<ec_dtype>(b)) != 0;
}

// ---------------------------------------------------------------- CellEncoding (src/encoding.rs:9-40)
template <typename T> struct CellEncoding;
#define EC_ENC(ID, P)                                             \
    template <> struct CellEncoding<P> {                          \
        static constexpr CellType cell_type() { return CellType::ID; } \
    };
EC_HOST_WITH_CT(EC_ENC)
#undef EC_ENC

// ---------------------------------------------------------------- CellValue (src/value.rs)
class CellValue {
    ec_value v_{};

public:
    CellValue() { v_.dtype = EC_U8; }
    explicit CellValue(const ec_value& v) : v_(v) {}
    template <typename T, typename = decltype(CellEncoding<T>::cell_type())>
    CellValue(T x) {  // CellValue::new / From<T> (value.rs:24-33, :111-115)
        std::memset(&v_, 0, sizeof v_);
        v_.dtype = static_cast<uint8_t>(CellEncoding<T>::cell_type());
        std::memcpy(&v_.v, &x, sizeof x);
    }
    const ec_value& raw() const { return v_; }
    CellType cell_type() const { return static_cast<CellType>(v_.dtype); }

    CellValue convert(CellType ct) const {  // value.rs:74-98
        ec_value out;
        check(ec_value_convert(&v_, static_cast<ec_dtype>(ct), &out));
        return CellValue(out);
    }
    template <typename T> T get() const {  // value.rs:51-67
        CellValue c = convert(CellEncoding<T>::cell_type());
        T x;
        std::memcpy(&x, &c.v_.v, sizeof x);
        return x;
    }
    std::pair<CellValue, CellValue> unify(const CellValue& o) const {  // value.rs:103-107
        CellType d = union_of(cell_type(), o.cell_type());
        return {convert(d), o.convert(d)};
    }
    double to_f64() const { return ec_value_to_f64(&v_); }  // value.rs:145-156
    uint64_t bits() const {
        uint64_t b = 0;
        std::memcpy(&b, &v_.v, size_of(cell_type()));
        return b;
    }
    static CellValue zero() { return CellValue(uint8_t(0)); }  // value.rs:166-170
    static CellValue one() { return CellValue(uint8_t(1)); }   // value.rs:159-164
    bool is_zero() const { return to_f64() == 0.0; }

    // impl Ord (value.rs:248-265): unify, ints natural, floats total_cmp
    int cmp(const CellValue& o) const {
        auto [l, r] = unify(o);
        auto key = [](const CellValue& c) -> int64_t {  // order-preserving int64 key of a unified value
            switch (c.cell_type()) {
                case CellType::UInt64: return static_cast<int64_t>(c.get<uint64_t>() ^ 0x8000000000000000ull);
                case CellType::Float32: {
                    int32_t b; float f = c.get<float>(); std::memcpy(&b, &f, 4);
                    return b ^ static_cast<int32_t>(static_cast<uint32_t>(b >> 31) >> 1);
                }
                case CellType::Float64: {
                    int64_t b; double d = c.get<double>(); std::memcpy(&b, &d, 8);
                    return b ^ static_cast<int64_t>(static_cast<uint64_t>(b >> 63) >> 1);
                }
                default: return is_signed(c.cell_type()) ? c.get<int64_t>() : static_cast<int64_t>(c.get<uint64_t>());
            }
        };
        int64_t a = key(l), b = key(r);
        return (a > b) - (a < b);
    }
    bool operator==(const CellValue& o) const { return cmp(o) == 0; }  // value.rs:267-271
    bool operator!=(const CellValue& o) const { return cmp(o) != 0; }
    bool operator<(const CellValue& o) const { return cmp(o) < 0; }
    bool operator>(const CellValue& o) const { return cmp(o) > 0; }
    CellValue min(const CellValue& o) const { return cmp(o) <= 0 ? *this : o; }
    CellValue max(const CellValue& o) const { return cmp(o) > 0 ? *this : o; }
};

// cv_bin_op! (value.rs:199-217): unify, to_f64 both, op -> always Float64. One scalar f64 op on the host.
#define EC_CV_OP(OPSYM)                                                   \
    inline CellValue operator OPSYM(const CellValue& l, const CellValue& r) { \
        auto [a, b] = l.unify(r);                                         \
        return CellValue(a.to_f64() OPSYM b.to_f64());                    \
    }
EC_CV_OP(+) EC_CV_OP(-) EC_CV_OP(*) EC_CV_OP(/)
#undef EC_CV_OP

inline CellValue operator-(const CellValue& v) {  // impl Neg (value.rs:224-240)
    switch (v.cell_type()) {
        case CellType::UInt8: return CellValue(static_cast<int16_t>(-static_cast<int16_t>(v.get<uint8_t>())));
        case CellType::UInt16: return CellValue(static_cast<int32_t>(-static_cast<int32_t>(v.get<uint16_t>())));
        case CellType::UInt32: return CellValue(-static_cast<double>(v.get<uint32_t>()));
        case CellType::UInt64: return CellValue(-static_cast<double>(v.get<uint64_t>()));
        case CellType::Int8: return CellValue(static_cast<int8_t>(0u - static_cast<uint8_t>(v.get<int8_t>())));
        case CellType::Int16: return CellValue(static_cast<int16_t>(0u - static_cast<uint16_t>(v.get<int16_t>())));
        case CellType::Int32: return CellValue(static_cast<int32_t>(0u - static_cast<uint32_t>(v.get<int32_t>())));
        case CellType::Int64: return CellValue(static_cast<int64_t>(0ull - static_cast<uint64_t>(v.get<int64_t>())));
        case CellType::Float32: return CellValue(-v.get<float>());
        default: return CellValue(-v.get<double>());
    }
}

inline CellValue zero(CellType ct) {  // CellType::zero (ctype.rs:134-143)
    switch (ct) {
#define EC_Z(ID, P) case CellType::ID: return CellValue(P(0));
        EC_HOST_WITH_CT(EC_Z)
#undef EC_Z
    }
    return CellValue();
}
inline CellValue one(CellType ct) {  // CellType::one (ctype.rs:146-155)
    switch (ct) {
#define EC_O(ID, P) case CellType::ID: return CellValue(P(1));
        EC_HOST_WITH_CT(EC_O)
#undef EC_O
    }
    return CellValue();
}
inline CellValue min_value(CellType ct) { ec_value v; check(ec_min_value(static_cast<ec_dtype>(ct), &v)); return CellValue(v); }
inline CellValue max_value(CellType ct) { ec_value v; check(ec_max_value(static_cast<ec_dtype>(ct), &v)); return CellValue(v); }

// ---------------------------------------------------------------- device plumbing
inline void init(int device = 0) { check(ec_init(device)); }

inline ec_stream& current_stream() { static thread_local ec_stream s = nullptr; return s; }

class DeviceMem {  // one HBM allocation, from the stream-ordered pool (no hipMalloc/hipFree per operator)
    void* p_ = nullptr;
    size_t bytes_ = 0;
    ec_stream alloc_stream_ = nullptr;

public:
    explicit DeviceMem(size_t bytes) : bytes_(bytes), alloc_stream_(current_stream()) { check(ec_alloc_async(&p_, bytes, alloc_stream_)); }
    // Back to the pool on the ALLOCATING stream, ordered after everything queued so far on the stream that is current
    // on the dropping thread (the buffer's last operator ran there if the caller switched streams in between, or if
    // another thread drops it): freeing on "whatever stream is current" alone would let the pool reuse the block
    // while kernels on the other stream still touch it.
    ~DeviceMem() { if (p_) ec_free_ordered(p_, alloc_stream_, current_stream()); }
    DeviceMem(const DeviceMem&) = delete;
    DeviceMem& operator=(const DeviceMem&) = delete;
    void* ptr() const { return p_; }
    size_t bytes() const { return bytes_; }
};

// ---------------------------------------------------------------- NoData<T> (src/masked/nodata.rs)
template <typename T>
struct NoData {
    enum Kind { NoneK, DefaultK, ValueK } kind = DefaultK;
    T v{};
    static NoData None() { return {NoneK, T{}}; }
    static NoData Default() { return {DefaultK, T{}}; }
    static NoData new_(T x) { return {ValueK, x}; }
    std::optional<T> value() const {  // nodata.rs:23-40
        if (kind == NoneK) return std::nullopt;
        if (kind == ValueK) return v;
        ec_value d;
        check(ec_nodata_default(static_cast<ec_dtype>(CellEncoding<T>::cell_type()), &d));
        T x;
        std::memcpy(&x, &d.v, sizeof x);
        return x;
    }
    bool is(const CellValue& value) const {  // nodata.rs:42-49
        auto nd = this->value();
        return nd ? CellValue(*nd) == value : false;
    }
};

// ---------------------------------------------------------------- CellBuffer (src/buffer.rs)
class CellBuffer {
    CellType ct_ = CellType::UInt8;
    size_t n_ = 0;
    std::shared_ptr<DeviceMem> mem_;

    static CellBuffer empty_u8() { return CellBuffer(CellType::UInt8, 0); }  // buffer.rs:233-234

public:
    CellBuffer() = default;
    CellBuffer(CellType ct, size_t n) : ct_(ct), n_(n), mem_(std::make_shared<DeviceMem>(n * size_of(ct))) {}
    CellBuffer(CellBuffer&&) = default;
    CellBuffer& operator=(CellBuffer&&) = default;
    CellBuffer(const CellBuffer&) = delete;  // Clone is explicit, as in Rust
    CellBuffer& operator=(const CellBuffer&) = delete;

    void* ptr() const { return mem_ ? mem_->ptr() : nullptr; }

    // ---- BufferOps (src/lib.rs:104-163)
    template <typename T> static CellBuffer from_vec(const std::vector<T>& data) {  // buffer.rs:64-66
        CellBuffer b(CellEncoding<T>::cell_type(), data.size());
        check(ec_upload(b.ptr(), data.data(), data.size() * sizeof(T), current_stream()));
        return b;
    }
    template <typename T> static CellBuffer new_(const std::vector<T>& data) { return from_vec(data); }
    // impl FromIterator<CellValue> (buffer.rs:229-250): empty -> UInt8; else the FIRST value's type, every
    // value through get::<T>().unwrap() (NarrowingError when one does not fit)
    static CellBuffer from_values(const std::vector<CellValue>& values) {
        if (values.empty()) return empty_u8();
        const CellType ct = values[0].cell_type();
        switch (ct) {
#define EC_FV(ID, P) case CellType::ID: { std::vector<P> v; for (const auto& x : values) v.push_back(x.get<P>()); return from_vec(v); }
            EC_HOST_WITH_CT(EC_FV)
#undef EC_FV
        }
        return empty_u8();
    }
    // impl IntoIterator for &CellBuffer (buffer.rs:278-305), as one download
    std::vector<CellValue> to_values() const {
        std::vector<CellValue> out;
        switch (ct_) {
#define EC_TV(ID, P) case CellType::ID: { for (P x : to_vec<P>()) out.emplace_back(x); break; }
            EC_HOST_WITH_CT(EC_TV)
#undef EC_TV
        }
        return out;
    }
    static CellBuffer with_defaults(size_t len, CellType ct) { return fill(len, zero(ct)); }  // buffer.rs:68-77
    static CellBuffer fill(size_t len, const CellValue& value) {                              // buffer.rs:79-88
        CellBuffer b(value.cell_type(), len);
        check(ec_fill(static_cast<ec_dtype>(value.cell_type()), b.ptr(), len, &value.raw(), current_stream()));
        return b;
    }
    template <typename T, typename F> static CellBuffer fill_via(size_t len, F f) {            // buffer.rs:90-97
        std::vector<T> v(len);
        for (size_t i = 0; i < len; ++i) v[i] = f(i);
        return from_vec(v);
    }
    size_t len() const { return n_; }
    bool is_empty() const { return n_ == 0; }
    CellType cell_type() const { return ct_; }
    // impl Debug for CellBuffer (buffer.rs:188-203); only the cells that are shown cross the bus
    std::string debug_string() const {
        std::vector<std::string> items;
        const size_t sz = size_of(ct_);
        auto fetch = [&](size_t off, size_t k) {
            std::vector<unsigned char> raw(k * sz);
            if (k) check(ec_download(raw.data(), static_cast<const char*>(ptr()) + off * sz, raw.size(), current_stream()));
            for (size_t i = 0; i < k; ++i) switch (ct_) {
#define EC_DBG(ID, P) case CellType::ID: { P v; std::memcpy(&v, raw.data() + i * sz, sz); items.push_back(rust_debug<P>(v)); break; }
                EC_HOST_WITH_CT(EC_DBG)
#undef EC_DBG
            }
        };
        if (n_ > 10) { fetch(0, 5); items.resize(11); fetch(n_ - 5, 5); }  // elided() drops the middle anyway
        else fetch(0, n_);
        return std::string(to_string(ct_)) + "CellBuffer(" + elided(items) + ")";
    }
    CellBuffer clone() const {
        CellBuffer b(ct_, n_);
        check(ec_copy(b.ptr(), ptr(), n_ * size_of(ct_), current_stream()));
        return b;
    }
    CellValue get(size_t index) const {  // buffer.rs:125-134; panics on OOB
        if (index >= n_) throw std::out_of_range("index out of bounds: the len is " + std::to_string(n_) + " but the index is " + std::to_string(index));
        ec_value v;
        std::memset(&v, 0, sizeof v);
        v.dtype = static_cast<uint8_t>(ct_);
        check(ec_download(&v.v, static_cast<const char*>(ptr()) + index * size_of(ct_), size_of(ct_), current_stream()));
        return CellValue(v);
    }
    void put(size_t idx, const CellValue& value) {  // buffer.rs:136-148
        CellValue c = value.convert(ct_);
        if (idx >= n_) throw std::out_of_range("index out of bounds");
        check(ec_upload(static_cast<char*>(ptr()) + idx * size_of(ct_), &c.raw().v, size_of(ct_), current_stream()));
    }
    // impl Extend<C> for CellBuffer (buffer.rs:205-221): every item through num-traits' range-checked
    // `to_<p>()` (value-based, unlike convert); a value that does not fit panics in the reference.
    template <typename C> void extend(const std::vector<C>& items) {
        const size_t sz = size_of(ct_), m = items.size();
        std::vector<unsigned char> bytes(m * sz);
        for (size_t i = 0; i < m; ++i) {
            const double f = static_cast<double>(items[i]);
            switch (ct_) {
#define EC_EXT(ID, P)                                                                                         \
    case CellType::ID: {                                                                                      \
        if (std::is_integral<P>::value) {                                                                     \
            const double lo = static_cast<double>(std::numeric_limits<P>::lowest()) - 1.0;                    \
            const double hi = static_cast<double>(std::numeric_limits<P>::max()) + 1.0;                       \
            if (f != f || !(f > lo && f < hi)) throw std::overflow_error("called `Option::unwrap()` on a `None` value"); \
        }                                                                                                     \
        const P v = static_cast<P>(items[i]);                                                                 \
        std::memcpy(bytes.data() + i * sz, &v, sz);                                                           \
        break;                                                                                                \
    }
                EC_HOST_WITH_CT(EC_EXT)
#undef EC_EXT
            }
        }
        CellBuffer grown(ct_, n_ + m);
        if (n_) check(ec_copy(grown.ptr(), ptr(), n_ * sz, current_stream()));
        if (m) check(ec_upload(static_cast<char*>(grown.ptr()) + n_ * sz, bytes.data(), bytes.size(), current_stream()));
        *this = std::move(grown);
    }
    CellBuffer convert(CellType cell_type) const {  // buffer.rs:150-167
        if (cell_type == ct_) return clone();
        if (!can_fit_into(ct_, cell_type)) throw NarrowingError(ct_, cell_type);
        if (n_ == 0) return empty_u8();
        CellBuffer out(cell_type, n_);
        check(ec_convert(static_cast<ec_dtype>(ct_), ptr(), static_cast<ec_dtype>(cell_type), out.ptr(), n_, current_stream()));
        return out;
    }
    std::pair<CellValue, CellValue> min_max() const {  // buffer.rs:169-173
        ec_value mn, mx;
        check(ec_min_max(static_cast<ec_dtype>(ct_), ptr(), nullptr, n_, &mn, &mx, current_stream()));
        return {CellValue(mn), CellValue(mx)};
    }
    template <typename T> std::vector<T> to_vec() const {  // buffer.rs:175-185
        CellBuffer r = convert(CellEncoding<T>::cell_type());
        if (r.cell_type() != CellEncoding<T>::cell_type())  // danger::cast assert (buffer.rs:444)
            throw std::logic_error("assertion failed: T::cell_type() == P::cell_type()");
        std::vector<T> out(r.len());
        check(ec_download(out.data(), r.ptr(), out.size() * sizeof(T), current_stream()));
        return out;
    }

    // ---- ops (buffer.rs:321-371)
    CellBuffer binop(ec_op op, const CellBuffer& rhs) const {
        size_t n = n_ < rhs.n_ ? n_ : rhs.n_;  // zip (buffer.rs:327)
        if (n == 0) return empty_u8();
        CellBuffer out(CellType::Float64, n);
        check(ec_binop(op, static_cast<ec_dtype>(ct_), ptr(), static_cast<ec_dtype>(rhs.ct_), rhs.ptr(), n,
                       static_cast<double*>(out.ptr()), current_stream()));
        return out;
    }
    CellBuffer binop(ec_op op, const CellValue& rhs) const {  // RHS scalar (buffer.rs:346-352)
        if (n_ == 0) return empty_u8();
        CellBuffer out(CellType::Float64, n_);
        check(ec_binop_scalar(op, static_cast<ec_dtype>(ct_), ptr(), n_, &rhs.raw(), static_cast<double*>(out.ptr()), current_stream()));
        return out;
    }
    CellBuffer neg() const {  // buffer.rs:360-365
        if (n_ == 0) return empty_u8();
        CellBuffer out(static_cast<CellType>(ec_neg_result_type(static_cast<ec_dtype>(ct_))), n_);
        check(ec_neg(static_cast<ec_dtype>(ct_), ptr(), n_, out.ptr(), current_stream()));
        return out;
    }

    // ---- Ord/Eq (buffer.rs:373-436): cell type first, then lexicographic total order, then length
    int cmp(const CellBuffer& o) const {  // decided on the device: first differing cell, no download
        int32_t res = 0;
        check(ec_buffer_cmp(static_cast<ec_dtype>(ct_), ptr(), n_, static_cast<ec_dtype>(o.ct_), o.ptr(), o.n_, &res, current_stream()));
        return res;
    }
    bool operator==(const CellBuffer& o) const { return cmp(o) == 0; }
    bool operator!=(const CellBuffer& o) const { return cmp(o) != 0; }
    bool operator<(const CellBuffer& o) const { return cmp(o) < 0; }
    bool operator>(const CellBuffer& o) const { return cmp(o) > 0; }
};

#define EC_CB_OP(OPSYM, OPC)                                                                            \
    inline CellBuffer operator OPSYM(const CellBuffer& l, const CellBuffer& r) { return l.binop(OPC, r); } \
    inline CellBuffer operator OPSYM(const CellBuffer& l, const CellValue& r) { return l.binop(OPC, r); }  \
    template <typename R, typename = decltype(CellEncoding<R>::cell_type())>                            \
    inline CellBuffer operator OPSYM(const CellBuffer& l, R r) { return l.binop(OPC, CellValue(r)); }
EC_CB_OP(+, EC_ADD) EC_CB_OP(-, EC_SUB) EC_CB_OP(*, EC_MUL) EC_CB_OP(/, EC_DIV)
#undef EC_CB_OP
inline CellBuffer operator-(const CellBuffer& b) { return b.neg(); }

// ---------------------------------------------------------------- Mask (src/masked/mask.rs)
class Mask {
    size_t n_ = 0;
    std::shared_ptr<DeviceMem> mem_;

public:
    Mask() : mem_(std::make_shared<DeviceMem>(0)) {}
    explicit Mask(size_t n) : n_(n), mem_(std::make_shared<DeviceMem>(n)) {}
    Mask(Mask&&) = default;
    Mask& operator=(Mask&&) = default;
    Mask(const Mask&) = delete;
    Mask& operator=(const Mask&) = delete;
    uint8_t* ptr() const { return static_cast<uint8_t*>(mem_->ptr()); }

    static Mask new_(const std::vector<bool>& values) {  // mask.rs:16-18
        std::vector<uint8_t> b(values.begin(), values.end());
        Mask m(b.size());
        check(ec_upload(m.ptr(), b.data(), b.size(), current_stream()));
        return m;
    }
    static Mask fill(size_t len, bool value) {  // mask.rs:21-23
        Mask m(len);
        CellValue v(static_cast<uint8_t>(value ? 1 : 0));
        check(ec_fill(EC_U8, m.ptr(), len, &v.raw(), current_stream()));
        return m;
    }
    template <typename F> static Mask fill_via(size_t len, F f) {  // mask.rs:28-33
        std::vector<bool> v(len);
        for (size_t i = 0; i < len; ++i) v[i] = f(i);
        return new_(v);
    }
    size_t len() const { return n_; }
    bool is_empty() const { return n_ == 0; }
    void extend(const std::vector<bool>& items) {  // impl Extend<bool> for Mask (mask.rs:83-87)
        std::vector<uint8_t> b(items.begin(), items.end());
        Mask grown(n_ + b.size());
        if (n_) check(ec_copy(grown.ptr(), ptr(), n_, current_stream()));
        if (!b.empty()) check(ec_upload(grown.ptr() + n_, b.data(), b.size(), current_stream()));
        *this = std::move(grown);
    }
    Mask clone() const {
        Mask m(n_);
        check(ec_copy(m.ptr(), ptr(), n_, current_stream()));
        return m;
    }
    void put(size_t index, bool value) {  // mask.rs:49-51
        if (index >= n_) throw std::out_of_range("index out of bounds");
        uint8_t b = value;
        check(ec_upload(ptr() + index, &b, 1, current_stream()));
    }
    bool get(size_t index) const {  // mask.rs:57-59
        if (index >= n_) throw std::out_of_range("index out of bounds");
        uint8_t b = 0;
        check(ec_download(&b, ptr() + index, 1, current_stream()));
        return b != 0;
    }
    std::pair<size_t, size_t> counts() const {  // mask.rs:72-80
        uint64_t t = 0, f = 0;
        check(ec_mask_counts(ptr(), n_, &t, &f, current_stream()));
        return {t, f};
    }
    bool all(bool value) const { auto [t, f] = counts(); return value ? f == 0 : t == 0; }  // mask.rs:67-69
    std::vector<bool> to_vec() const {
        std::vector<uint8_t> b(n_);
        check(ec_download(b.data(), ptr(), n_, current_stream()));
        return std::vector<bool>(b.begin(), b.end());
    }
    int cmp(const Mask& o) const {  // derived Ord on Vec<bool> (mask.rs:10)
        int32_t res = 0;
        check(ec_buffer_cmp(EC_U8, ptr(), n_, EC_U8, o.ptr(), o.n_, &res, current_stream()));
        return res;
    }
    bool operator==(const Mask& o) const { return cmp(o) == 0; }
    bool operator!=(const Mask& o) const { return !(*this == o); }
    std::string debug_string() const {  // impl Debug for Mask (mask.rs:165-169)
        std::vector<std::string> items;
        auto fetch = [&](size_t off, size_t k) {
            std::vector<uint8_t> raw(k);
            if (k) check(ec_download(raw.data(), ptr() + off, k, current_stream()));
            for (uint8_t b : raw) items.push_back(rust_debug<bool>(b != 0));
        };
        if (n_ > 10) { fetch(0, 5); items.resize(11); fetch(n_ - 5, 5); }
        else fetch(0, n_);
        return "Mask(" + elided(items) + ")";
    }

    Mask operator!() const& {  // Not for &Mask (mask.rs:111-116)
        Mask m(n_);
        check(ec_mask_not(ptr(), n_, m.ptr(), current_stream()));
        return m;
    }
    Mask operator!() && {  // Not for Mask: in place (mask.rs:103-109)
        check(ec_mask_not(ptr(), n_, ptr(), current_stream()));
        return std::move(*this);
    }
};
inline Mask operator&(const Mask& l, const Mask& r) {  // BitAnd for &Mask: zip -> shorter (mask.rs:129-140)
    size_t n = l.len() < r.len() ? l.len() : r.len();
    Mask m(n);
    check(ec_mask_and(l.ptr(), r.ptr(), n, m.ptr(), current_stream()));
    return m;
}
inline Mask operator|(const Mask& l, const Mask& r) {  // BitOr for &Mask (mask.rs:153-163)
    size_t n = l.len() < r.len() ? l.len() : r.len();
    Mask m(n);
    check(ec_mask_or(l.ptr(), r.ptr(), n, m.ptr(), current_stream()));
    return m;
}
inline Mask operator&(Mask&& l, const Mask& r) {  // BitAnd for Mask (owned, in place, lhs length kept — mask.rs:118-127)
    check(ec_mask_and(l.ptr(), r.ptr(), l.len() < r.len() ? l.len() : r.len(), l.ptr(), current_stream()));
    return std::move(l);
}
inline Mask operator|(Mask&& l, const Mask& r) {  // mask.rs:142-151
    check(ec_mask_or(l.ptr(), r.ptr(), l.len() < r.len() ? l.len() : r.len(), l.ptr(), current_stream()));
    return std::move(l);
}

// ---------------------------------------------------------------- MaskedCellBuffer (src/masked/masked_buffer.rs)
class MaskedCellBuffer {
    CellBuffer buf_;
    Mask mask_;

public:
    MaskedCellBuffer(CellBuffer buffer, Mask mask) : buf_(std::move(buffer)), mask_(std::move(mask)) {  // :48-55
        if (buf_.len() != mask_.len()) throw std::logic_error("Mask and buffer must have the same length.");
    }
    static MaskedCellBuffer new_(CellBuffer b, Mask m) { return MaskedCellBuffer(std::move(b), std::move(m)); }
    MaskedCellBuffer(MaskedCellBuffer&&) = default;
    MaskedCellBuffer& operator=(MaskedCellBuffer&&) = default;
    MaskedCellBuffer clone() const { return MaskedCellBuffer(buf_.clone(), mask_.clone()); }

    template <typename T> static MaskedCellBuffer from_vec(const std::vector<T>& data) {  // :156-160
        CellBuffer b = CellBuffer::from_vec(data);
        size_t n = b.len();
        return MaskedCellBuffer(std::move(b), Mask::fill(n, true));
    }
    static MaskedCellBuffer from(CellBuffer b) {  // From<CellBuffer> (:250-255)
        size_t n = b.len();
        return MaskedCellBuffer(std::move(b), Mask::fill(n, true));
    }
    template <typename T> static MaskedCellBuffer from_vec_with_nodata(const std::vector<T>& data, NoData<T> nodata) {  // :62-71
        CellBuffer b = CellBuffer::from_vec(data);
        Mask m(b.len());
        auto nd = nodata.value();
        CellValue ndv = nd ? CellValue(*nd) : CellValue();
        check(ec_mask_from_nodata(static_cast<ec_dtype>(b.cell_type()), b.ptr(), b.len(), nd ? &ndv.raw() : nullptr, m.ptr(), current_stream()));
        return MaskedCellBuffer(std::move(b), std::move(m));
    }
    static MaskedCellBuffer with_defaults(size_t len, CellType ct) { return from(CellBuffer::with_defaults(len, ct)); }
    static MaskedCellBuffer fill(size_t len, const CellValue& v) { return from(CellBuffer::fill(len, v)); }
    template <typename T, typename F> static MaskedCellBuffer fill_via(size_t len, F f) { return from(CellBuffer::fill_via<T>(len, f)); }
    template <typename T, typename F> static MaskedCellBuffer fill_with_mask_via(size_t len, F mv) {  // :73-79
        std::vector<T> v(len);
        std::vector<bool> m(len);
        for (size_t i = 0; i < len; ++i) { auto p = mv(i); v[i] = p.first; m[i] = p.second; }
        return MaskedCellBuffer(CellBuffer::from_vec(v), Mask::new_(m));
    }

    const CellBuffer& buffer() const { return buf_; }
    CellBuffer& buffer_mut() { return buf_; }
    const Mask& mask() const { return mask_; }
    Mask& mask_mut() { return mask_; }
    size_t len() const { return buf_.len(); }
    CellType cell_type() const { return buf_.cell_type(); }
    CellValue get(size_t i) const { return buf_.get(i); }
    void put(size_t i, const CellValue& v) { buf_.put(i, v); }
    std::optional<CellValue> get_masked(size_t i) const {  // :100-106
        if (mask_.get(i)) return buf_.get(i);
        return std::nullopt;
    }
    std::pair<CellValue, bool> get_with_mask(size_t i) const { return {buf_.get(i), mask_.get(i)}; }
    void put_with_mask(size_t i, const CellValue& v, bool m) { put(i, v); mask_.put(i, m); }  // :120-129
    std::pair<size_t, size_t> counts() const { return mask_.counts(); }                       // :132-134
    template <typename C> void extend(const std::vector<std::pair<C, bool>>& items) {         // :280-287
        std::vector<C> v;
        std::vector<bool> m;
        for (const auto& p : items) { v.push_back(p.first); m.push_back(p.second); }
        buf_.extend(v);
        mask_.extend(m);
    }
    MaskedCellBuffer convert(CellType ct) const { return MaskedCellBuffer(buf_.convert(ct), mask_.clone()); }  // :200-206
    template <typename T> std::vector<T> to_vec() const { return buf_.to_vec<T>(); }
    template <typename T> std::vector<T> to_vec_with_nodata(NoData<T> no_data) const {  // :137-152
        CellBuffer conv = buf_.convert(CellEncoding<T>::cell_type());
        if (conv.cell_type() != CellEncoding<T>::cell_type()) throw std::logic_error("assertion failed: T::cell_type() == P::cell_type()");
        auto nd = no_data.value();
        std::vector<T> out(conv.len());
        if (!nd) {
            check(ec_download(out.data(), conv.ptr(), out.size() * sizeof(T), current_stream()));
            return out;
        }
        CellBuffer sel(conv.cell_type(), conv.len());
        CellValue ndv(*nd);
        check(ec_mask_select(static_cast<ec_dtype>(conv.cell_type()), conv.ptr(), mask_.ptr(), conv.len(), &ndv.raw(), sel.ptr(), current_stream()));
        check(ec_download(out.data(), sel.ptr(), out.size() * sizeof(T), current_stream()));
        return out;
    }
    std::pair<CellValue, CellValue> min_max() const {  // :208-217
        ec_value mn, mx;
        check(ec_min_max(static_cast<ec_dtype>(cell_type()), buf_.ptr(), mask_.ptr(), len(), &mn, &mx, current_stream()));
        return {CellValue(mn), CellValue(mx)};
    }

    // ---- ops (:323-383)
    MaskedCellBuffer binop(ec_op op, const MaskedCellBuffer& rhs) const {
        size_t n = len() < rhs.len() ? len() : rhs.len();
        if (n == 0) return MaskedCellBuffer(CellBuffer(CellType::UInt8, 0), Mask(0));
        CellBuffer out(CellType::Float64, n);
        Mask om(n);
        check(ec_masked_binop(op, static_cast<ec_dtype>(cell_type()), buf_.ptr(), mask_.ptr(), static_cast<ec_dtype>(rhs.cell_type()),
                              rhs.buf_.ptr(), rhs.mask_.ptr(), n, static_cast<double*>(out.ptr()), om.ptr(), current_stream()));
        return MaskedCellBuffer(std::move(out), std::move(om));
    }
    MaskedCellBuffer binop(ec_op op, const CellValue& rhs) const { return MaskedCellBuffer(buf_.binop(op, rhs), mask_.clone()); }
    MaskedCellBuffer neg() const { return MaskedCellBuffer(buf_.neg(), mask_.clone()); }
    bool operator==(const MaskedCellBuffer& o) const { return buf_ == o.buf_ && mask_ == o.mask_; }  // derived PartialEq (:39)
    std::string debug_string() const {  // debug_tuple(buffer, mask) (:227-235)
        return std::string(to_string(cell_type())) + "MaskedCellBuffer(" + buf_.debug_string() + ", " + mask_.debug_string() + ")";
    }
    int cmp(const MaskedCellBuffer& o) const {  // derived PartialOrd (:39): buffer first, the mask breaks ties
        const int c = buf_.cmp(o.buf_);
        return c != 0 ? c : mask_.cmp(o.mask_);
    }
    bool operator<(const MaskedCellBuffer& o) const { return cmp(o) < 0; }
    bool operator>(const MaskedCellBuffer& o) const { return cmp(o) > 0; }
    bool operator!=(const MaskedCellBuffer& o) const { return !(*this == o); }
};

#define EC_MCB_OP(OPSYM, OPC)                                                                                          \
    inline MaskedCellBuffer operator OPSYM(const MaskedCellBuffer& l, const MaskedCellBuffer& r) { return l.binop(OPC, r); } \
    inline MaskedCellBuffer operator OPSYM(const MaskedCellBuffer& l, const CellValue& r) { return l.binop(OPC, r); }        \
    template <typename R, typename = decltype(CellEncoding<R>::cell_type())>                                            \
    inline MaskedCellBuffer operator OPSYM(const MaskedCellBuffer& l, R r) { return l.binop(OPC, CellValue(r)); }
EC_MCB_OP(+, EC_ADD) EC_MCB_OP(-, EC_SUB) EC_MCB_OP(*, EC_MUL) EC_MCB_OP(/, EC_DIV)
#undef EC_MCB_OP
inline MaskedCellBuffer operator-(const MaskedCellBuffer& b) { return b.neg(); }

// ---------------------------------------------------------------- fused operator chains
// The reference evaluates `(&nir - &red) / (nir + red)` (src/gdal/rasterband.rs:148) eagerly, one pass and
// one f64 temporary per operator.  `lazy(buf)` wraps a buffer so that the same operator syntax builds a
// two-level expression which `eval()` runs as ONE kernel (ec_fused); every step is the same rounded f64
// op, so the result is bit-identical to the eager chain.
namespace fused {

inline CellBuffer expr(const CellBuffer& x, ec_op o1, const CellBuffer& y, ec_op o2, const CellBuffer& z,
                       ec_op o3 = EC_OP_NONE, const CellBuffer* w = nullptr) {
    size_t n = std::min(std::min(x.len(), y.len()), z.len());
    if (o3 != EC_OP_NONE) n = std::min(n, w->len());  // zip truncation of every step (buffer.rs:327)
    if (n == 0) return CellBuffer(CellType::UInt8, 0);
    const ec_dtype dt[4] = {static_cast<ec_dtype>(x.cell_type()), static_cast<ec_dtype>(y.cell_type()),
                            static_cast<ec_dtype>(z.cell_type()), static_cast<ec_dtype>(w ? w->cell_type() : z.cell_type())};
    const void* p[4] = {x.ptr(), y.ptr(), z.ptr(), w ? w->ptr() : nullptr};
    CellBuffer out(CellType::Float64, n);
    check(ec_fused(o1, o2, o3, dt, p, nullptr, n, static_cast<double*>(out.ptr()), current_stream()));
    return out;
}
// (x o1 y) o2 scalar — the RHS-scalar operator form (src/buffer.rs:346-352), e.g. `(buf + ones) * 2.0`
inline CellBuffer expr(const CellBuffer& x, ec_op o1, const CellBuffer& y, ec_op o2, const CellValue& z) {
    const size_t n = std::min(x.len(), y.len());
    if (n == 0) return CellBuffer(CellType::UInt8, 0);
    const ec_dtype dt[4] = {static_cast<ec_dtype>(x.cell_type()), static_cast<ec_dtype>(y.cell_type()), 0, 0};
    const void* p[4] = {x.ptr(), y.ptr(), nullptr, nullptr};
    ec_value sc[4] = {};
    sc[2] = z.raw();
    CellBuffer out(CellType::Float64, n);
    check(ec_fused(o1, o2, EC_OP_NONE, dt, p, sc, n, static_cast<double*>(out.ptr()), current_stream()));
    return out;
}
inline MaskedCellBuffer expr(const MaskedCellBuffer& x, ec_op o1, const MaskedCellBuffer& y, ec_op o2, const MaskedCellBuffer& z,
                             ec_op o3 = EC_OP_NONE, const MaskedCellBuffer* w = nullptr) {
    size_t n = std::min(std::min(x.len(), y.len()), z.len());
    if (o3 != EC_OP_NONE) n = std::min(n, w->len());
    if (n == 0) return MaskedCellBuffer(CellBuffer(CellType::UInt8, 0), Mask(0));
    const ec_dtype dt[4] = {static_cast<ec_dtype>(x.cell_type()), static_cast<ec_dtype>(y.cell_type()),
                            static_cast<ec_dtype>(z.cell_type()), static_cast<ec_dtype>(w ? w->cell_type() : z.cell_type())};
    const void* p[4] = {x.buffer().ptr(), y.buffer().ptr(), z.buffer().ptr(), w ? w->buffer().ptr() : nullptr};
    const uint8_t* m[4] = {x.mask().ptr(), y.mask().ptr(), z.mask().ptr(), w ? w->mask().ptr() : nullptr};
    CellBuffer out(CellType::Float64, n);
    Mask om(n);
    check(ec_masked_fused(o1, o2, o3, dt, p, m, nullptr, n, static_cast<double*>(out.ptr()), om.ptr(), current_stream()));
    return MaskedCellBuffer(std::move(out), std::move(om));
}
inline MaskedCellBuffer expr(const MaskedCellBuffer& x, ec_op o1, const MaskedCellBuffer& y, ec_op o2, const CellValue& z) {
    const size_t n = std::min(x.len(), y.len());
    if (n == 0) return MaskedCellBuffer(CellBuffer(CellType::UInt8, 0), Mask(0));
    const ec_dtype dt[4] = {static_cast<ec_dtype>(x.cell_type()), static_cast<ec_dtype>(y.cell_type()), 0, 0};
    const void* p[4] = {x.buffer().ptr(), y.buffer().ptr(), nullptr, nullptr};
    const uint8_t* m[4] = {x.mask().ptr(), y.mask().ptr(), nullptr, nullptr};
    ec_value sc[4] = {};
    sc[2] = z.raw();
    CellBuffer out(CellType::Float64, n);
    Mask om(n);
    check(ec_masked_fused(o1, o2, EC_OP_NONE, dt, p, m, sc, n, static_cast<double*>(out.ptr()), om.ptr(), current_stream()));
    return MaskedCellBuffer(std::move(out), std::move(om));
}
template <typename B> inline B ndvi(const B& nir, const B& red) { return expr(nir, EC_SUB, red, EC_DIV, nir, EC_ADD, &red); }

// expression-template front end: lazy(a) - lazy(b) etc.
template <typename B> struct Leaf { const B* b; };
template <typename B> struct Node { const B* l; const B* r; ec_op op; };                                  // leaf op leaf
template <typename B> struct Tree { Node<B> l; ec_op op; const B* rl; const B* rr; ec_op rop; };           // node op (leaf | node)
template <typename B> inline Leaf<B> lazy(const B& b) { return Leaf<B>{&b}; }
#define EC_LAZY_OP(SYM, OPC)                                                                                             \
    template <typename B> inline Node<B> operator SYM(Leaf<B> a, Leaf<B> b) { return Node<B>{a.b, b.b, OPC}; }           \
    template <typename B> inline Tree<B> operator SYM(Node<B> a, Leaf<B> b) { return Tree<B>{a, OPC, b.b, nullptr, EC_OP_NONE}; } \
    template <typename B> inline Tree<B> operator SYM(Node<B> a, Node<B> b) { return Tree<B>{a, OPC, b.l, b.r, b.op}; }
EC_LAZY_OP(+, EC_ADD) EC_LAZY_OP(-, EC_SUB) EC_LAZY_OP(*, EC_MUL) EC_LAZY_OP(/, EC_DIV)
#undef EC_LAZY_OP
template <typename B> struct ScalarTree { Node<B> l; ec_op op; CellValue s; };                              // node op scalar
#define EC_LAZY_SC(SYM, OPC)                                                                                             \
    template <typename B> inline ScalarTree<B> operator SYM(Node<B> a, const CellValue& s) { return ScalarTree<B>{a, OPC, s}; } \
    template <typename B, typename R, typename = decltype(CellEncoding<R>::cell_type())>                                 \
    inline ScalarTree<B> operator SYM(Node<B> a, R s) { return ScalarTree<B>{a, OPC, CellValue(s)}; }
EC_LAZY_SC(+, EC_ADD) EC_LAZY_SC(-, EC_SUB) EC_LAZY_SC(*, EC_MUL) EC_LAZY_SC(/, EC_DIV)
#undef EC_LAZY_SC
template <typename B> inline B eval(const ScalarTree<B>& t) { return expr(*t.l.l, t.l.op, *t.l.r, t.op, t.s); }
template <typename B> inline B eval(const Node<B>& n) { return n.l->binop(n.op, *n.r); }
template <typename B> inline B eval(const Tree<B>& t) { return expr(*t.l.l, t.l.op, *t.l.r, t.op, *t.rl, t.rop, t.rr); }


// An operator tree of ANY depth in one pass (ec_expr): up to four buffers of any cell types, eight scalars, sixteen steps
// `reg[dst] = a op b` over four f64 registers; the value is what the last step computed.  Same bits as the eager
// evaluation of the same operators in the same order.  EVI, 2.5 (nir - red) / (nir + 6 red - 7.5 blue + 1):
//   using namespace ec::fused;
//   auto evi = program({&nir, &red, &blue}, {2.5, 6.0, 7.5, 1.0},
//                      {{EC_SUB, stream(0), stream(1), 0}, {EC_MUL, reg(0), scalar(0), 0}, {EC_MUL, stream(1), scalar(1), 1},
//                       {EC_ADD, stream(0), reg(1), 1},    {EC_MUL, stream(2), scalar(2), 2}, {EC_SUB, reg(1), reg(2), 1},
//                       {EC_ADD, reg(1), scalar(3), 1},    {EC_DIV, reg(0), reg(1), 0}});
// A malformed program (bad reference, register read before written, counts out of range) throws Error (EC_ERR_ARG).
constexpr int8_t stream(int k) { return EC_EXPR_STREAM(k); }
constexpr int8_t reg(int k) { return EC_EXPR_REG(k); }
constexpr int8_t scalar(int k) { return EC_EXPR_SCALAR(k); }
inline CellBuffer program(const std::vector<const CellBuffer*>& streams, const std::vector<CellValue>& scalars,
                          const std::vector<ec_expr_step>& steps) {
    size_t n = streams.empty() ? 0 : streams[0]->len();
    std::vector<ec_dtype> dt;
    std::vector<const void*> p;
    for (const CellBuffer* b : streams) {
        n = std::min(n, b->len());
        dt.push_back(static_cast<ec_dtype>(b->cell_type()));
        p.push_back(b->ptr());
    }
    if (n == 0) return CellBuffer(CellType::UInt8, 0);
    std::vector<ec_value> sc;
    for (const CellValue& v : scalars) sc.push_back(v.raw());
    CellBuffer out(CellType::Float64, n);
    check(ec_expr(dt.data(), p.data(), static_cast<int32_t>(streams.size()), sc.data(), static_cast<int32_t>(sc.size()), steps.data(),
                  static_cast<int32_t>(steps.size()), n, static_cast<double*>(out.ptr()), current_stream()));
    return out;
}
inline MaskedCellBuffer program(const std::vector<const MaskedCellBuffer*>& streams, const std::vector<CellValue>& scalars,
                                const std::vector<ec_expr_step>& steps) {
    size_t n = streams.empty() ? 0 : streams[0]->len();
    std::vector<ec_dtype> dt;
    std::vector<const void*> p;
    std::vector<const uint8_t*> m;
    for (const MaskedCellBuffer* b : streams) {
        n = std::min(n, b->len());
        dt.push_back(static_cast<ec_dtype>(b->cell_type()));
        p.push_back(b->buffer().ptr());
        m.push_back(b->mask().ptr());
    }
    if (n == 0) return MaskedCellBuffer(CellBuffer(CellType::UInt8, 0), Mask(0));
    std::vector<ec_value> sc;
    for (const CellValue& v : scalars) sc.push_back(v.raw());
    CellBuffer out(CellType::Float64, n);
    Mask om(n);
    check(ec_masked_expr(dt.data(), p.data(), m.data(), static_cast<int32_t>(streams.size()), sc.data(), static_cast<int32_t>(sc.size()),
                         steps.data(), static_cast<int32_t>(steps.size()), n, static_cast<double*>(out.ptr()), om.ptr(), current_stream()));
    return MaskedCellBuffer(std::move(out), std::move(om));
}


// (min, max) of a program's result without its raster (ec_expr_min_max): once the library has compiled the program for itself
// only the streams are read; until then it runs the program into a temporary and reduces that.  Typed Float64.
inline std::pair<CellValue, CellValue> program_min_max(const std::vector<const CellBuffer*>& streams, const std::vector<CellValue>& scalars,
                                                      const std::vector<ec_expr_step>& steps) {
    size_t n = streams.empty() ? 0 : streams[0]->len();
    std::vector<ec_dtype> dt;
    std::vector<const void*> p;
    for (const CellBuffer* b : streams) {
        n = std::min(n, b->len());
        dt.push_back(static_cast<ec_dtype>(b->cell_type()));
        p.push_back(b->ptr());
    }
    std::vector<ec_value> sc;
    for (const CellValue& v : scalars) sc.push_back(v.raw());
    ec_value mn{}, mx{};
    check(ec_expr_min_max(dt.data(), p.data(), nullptr, static_cast<int32_t>(streams.size()), sc.data(), static_cast<int32_t>(sc.size()),
                          steps.data(), static_cast<int32_t>(steps.size()), n, &mn, &mx, current_stream()));
    return {CellValue(mn), CellValue(mx)};
}
inline std::pair<CellValue, CellValue> program_min_max(const std::vector<const MaskedCellBuffer*>& streams, const std::vector<CellValue>& scalars,
                                                      const std::vector<ec_expr_step>& steps) {
    size_t n = streams.empty() ? 0 : streams[0]->len();
    std::vector<ec_dtype> dt;
    std::vector<const void*> p;
    std::vector<const uint8_t*> m;
    for (const MaskedCellBuffer* b : streams) {
        n = std::min(n, b->len());
        dt.push_back(static_cast<ec_dtype>(b->cell_type()));
        p.push_back(b->buffer().ptr());
        m.push_back(b->mask().ptr());
    }
    std::vector<ec_value> sc;
    for (const CellValue& v : scalars) sc.push_back(v.raw());
    ec_value mn{}, mx{};
    check(ec_expr_min_max(dt.data(), p.data(), m.data(), static_cast<int32_t>(streams.size()), sc.data(), static_cast<int32_t>(sc.size()),
                          steps.data(), static_cast<int32_t>(steps.size()), n, &mn, &mx, current_stream()));
    return {CellValue(mn), CellValue(mx)};
}

// Host memory in, host memory out (ec_host_expr): the same program over HOST arrays of n cells each — `arrays[k]` is
// {cell type, pointer} — streamed through the GPU in chunks with upload, kernel and download overlapped; PCIe-bound
// (≈ 6 Gcells/s at 16384² against ≈ 1.2 for from_vec + operator + to_vec).  Page-locked buffers (ec_host_alloc) are copied
// as they are, others are registered for the duration of the call.
inline std::vector<double> program_host(const std::vector<std::pair<CellType, const void*>>& arrays, size_t n,
                                        const std::vector<CellValue>& scalars, const std::vector<ec_expr_step>& steps, size_t chunk_cells = 0) {
    std::vector<ec_dtype> dt;
    std::vector<const void*> p;
    for (const auto& a : arrays) {
        dt.push_back(static_cast<ec_dtype>(a.first));
        p.push_back(a.second);
    }
    std::vector<ec_value> sc;
    for (const CellValue& v : scalars) sc.push_back(v.raw());
    std::vector<double> out(n);
    check(ec_host_expr(dt.data(), p.data(), static_cast<int32_t>(arrays.size()), sc.data(), static_cast<int32_t>(sc.size()), steps.data(),
                       static_cast<int32_t>(steps.size()), n, out.data(), chunk_cells));
    return out;
}

// Operator syntax for trees of ANY depth: `tree(nir)` wraps a buffer so that + - * / build a run-time operator tree, with
// buffers (all plain or all masked) and scalars as leaves on either side; `eval()` schedules it onto the four registers
// of an expression program — post-order, the sub-tree that needs more registers first (Sethi-Ullman), a register freed as
// soon as its value is consumed — and runs it as ONE ec_expr launch.  A tree that needs more than 4 distinct buffers, 8
// scalars, 16 operators or 4 live temporaries is cut: its two sub-trees are evaluated first (each again as far as it
// goes) and combined with one operator.  Bit-identical to the same operators applied eagerly, in the same order.
//   auto evi = ((tree(nir) - red) * 2.5 / (tree(nir) + tree(red) * 6.0 - tree(blue) * 7.5 + 1.0)).eval();
template <typename B>
class Expr {
    struct Node {
        ec_op op = EC_OP_NONE;  // EC_OP_NONE: a leaf
        std::shared_ptr<const Node> l, r;
        const B* buf = nullptr;  // leaf: a buffer, or
        CellValue scalar;        //       a scalar
        std::shared_ptr<const B> owned;  // a temporary that a cut produced (kept alive by the tree)
    };
    std::shared_ptr<const Node> n_;
    explicit Expr(std::shared_ptr<const Node> n) : n_(std::move(n)) {}
    static std::shared_ptr<const Node> leaf(const B& b) { auto n = std::make_shared<Node>(); n->buf = &b; return n; }
    static std::shared_ptr<const Node> leaf(std::shared_ptr<const B> b) { auto n = std::make_shared<Node>(); n->buf = b.get(); n->owned = std::move(b); return n; }

    struct Compiler {
        std::vector<const B*> streams;
        std::vector<CellValue> scalars;
        std::vector<ec_expr_step> steps;
        bool free_[4] = {true, true, true, true};
        bool overflow = false;
        static int need(const Node& t) {
            if (t.op == EC_OP_NONE) return 0;
            const int l = need(*t.l), r = need(*t.r);
            return l != r ? std::max(1, std::max(l, r)) : (l ? l + 1 : 1);
        }
        int8_t operand(const Node& t) {
            if (t.op != EC_OP_NONE) return emit(t);
            if (t.buf) {
                for (size_t i = 0; i < streams.size(); ++i)
                    if (streams[i] == t.buf) return stream(int(i));
                if (streams.size() == EC_EXPR_MAX_STREAMS) { overflow = true; return 0; }
                streams.push_back(t.buf);
                return stream(int(streams.size()) - 1);
            }
            if (scalars.size() == EC_EXPR_MAX_SCALARS) { overflow = true; return 0; }
            scalars.push_back(t.scalar);
            return fused::scalar(int(scalars.size()) - 1);
        }
        int8_t emit(const Node& t) {
            int8_t a, b;
            if (need(*t.r) > need(*t.l)) { b = operand(*t.r); a = operand(*t.l); }
            else { a = operand(*t.l); b = operand(*t.r); }
            if (overflow) return 0;
            for (int8_t ref : {a, b})  // operands that are registers are dead after this step
                if (ref >= reg(0) && ref < fused::scalar(0)) free_[ref - reg(0)] = true;
            int dst = -1;
            for (int k = 0; k < 4 && dst < 0; ++k) if (free_[k]) dst = k;
            if (dst < 0 || steps.size() == EC_EXPR_MAX_STEPS) { overflow = true; return 0; }
            free_[dst] = false;
            steps.push_back(ec_expr_step{static_cast<int8_t>(t.op), a, b, static_cast<int8_t>(dst)});
            return reg(dst);
        }
    };
    static bool has_buffer(const Node& t) { return t.op == EC_OP_NONE ? t.buf != nullptr : has_buffer(*t.l) || has_buffer(*t.r); }
    static CellValue fold(const Node& t) {  // a sub-tree of scalars only: host arithmetic (impl $trt for CellValue, src/value.rs:207)
        if (t.op == EC_OP_NONE) return t.scalar;
        const CellValue a = fold(*t.l), b = fold(*t.r);
        return t.op == EC_ADD ? a + b : t.op == EC_SUB ? a - b : t.op == EC_MUL ? a * b : a / b;
    }
    static B run(const Node& t) {
        if (t.op == EC_OP_NONE) throw Error(EC_ERR_ARG, "fused::Expr::eval: the tree is a single leaf");
        Compiler c;
        c.emit(t);
        if (!c.overflow && !c.streams.empty()) return program(c.streams, c.scalars, c.steps);
        // cut: both sub-trees first, then one operator
        auto side = [](const std::shared_ptr<const Node>& s) -> std::shared_ptr<const Node> {
            if (s->op == EC_OP_NONE) return s;
            if (!has_buffer(*s)) { auto n = std::make_shared<Node>(); n->scalar = fold(*s); return n; }
            return leaf(std::make_shared<const B>(run(*s)));
        };
        Node top;
        top.op = t.op;
        top.l = side(t.l);
        top.r = side(t.r);
        Compiler two;
        two.emit(top);
        if (two.overflow || two.streams.empty()) throw Error(EC_ERR_ARG, "fused::Expr::eval: the tree has no buffer operand");
        return program(two.streams, two.scalars, two.steps);
    }
    static Expr node(ec_op op, const Expr& a, const Expr& b) {
        auto n = std::make_shared<Node>();
        n->op = op; n->l = a.n_; n->r = b.n_;
        return Expr(std::move(n));
    }

public:
    Expr(const B& b) : n_(leaf(b)) {}  // the buffer must outlive the expression
    Expr(const CellValue& v) { auto n = std::make_shared<Node>(); n->scalar = v; n_ = std::move(n); }
    template <typename T, typename = decltype(CellEncoding<T>::cell_type())>
    Expr(T v) : Expr(CellValue(v)) {}
    B eval() const { return run(*n_); }
    // (min, max) of the tree's result without its raster when the tree fits one expression program (ec_expr_min_max);
    // otherwise eval().min_max()
    std::pair<CellValue, CellValue> min_max() const {
        if (n_->op != EC_OP_NONE) {
            Compiler c;
            c.emit(*n_);
            if (!c.overflow && !c.streams.empty()) return program_min_max(c.streams, c.scalars, c.steps);
        }
        return eval().min_max();
    }
#define EC_EXPR_OP(SYM, OPC)                                                                             \
    friend Expr operator SYM(const Expr& a, const Expr& b) { return node(OPC, a, b); }                   \
    friend Expr operator SYM(const Expr& a, const B& b) { return node(OPC, a, Expr(b)); }                \
    friend Expr operator SYM(const B& a, const Expr& b) { return node(OPC, Expr(a), b); }                \
    template <typename T, typename = decltype(CellEncoding<T>::cell_type())>                             \
    friend Expr operator SYM(const Expr& a, T b) { return node(OPC, a, Expr(CellValue(b))); }            \
    template <typename T, typename = decltype(CellEncoding<T>::cell_type())>                             \
    friend Expr operator SYM(T a, const Expr& b) { return node(OPC, Expr(CellValue(a)), b); }
    EC_EXPR_OP(+, EC_ADD) EC_EXPR_OP(-, EC_SUB) EC_EXPR_OP(*, EC_MUL) EC_EXPR_OP(/, EC_DIV)
#undef EC_EXPR_OP
};
template <typename B> inline Expr<B> tree(const B& b) { return Expr<B>(b); }

}  // namespace fused

// ---------------------------------------------------------------- one process, all GPUs of the node
// Row-block shards over an ec_shard_group: shard i on device i, a launch thread + stream + RCCL communicator per
// device inside the library.  Element-wise ops are queued for the launch threads and return at once (fire-and-forget:
// a failure inside one is reported by the next sync() / reduction); min_max / counts all-reduce their 16-byte payloads
// over xGMI and wait for the result.  (One process per GPU instead: ec_comm_init_rank + ec_allreduce_*.)
namespace sharded {

class ShardGroup {
    ec_shard_group* g_ = nullptr;
    int n_ = 0;

public:
    explicit ShardGroup(const std::vector<int32_t>& devices, uint32_t flags = EC_GROUP_RCCL) : n_(static_cast<int>(devices.size())) {
        check(ec_shard_group_create(devices.data(), n_, flags, &g_));
    }
    ~ShardGroup() { ec_shard_group_destroy(g_); }
    ShardGroup(const ShardGroup&) = delete;
    ShardGroup& operator=(const ShardGroup&) = delete;
    ec_shard_group* raw() const { return g_; }
    int size() const { return n_; }
    // waits for everything queued so far; throws the first failure a fire-and-forget call left behind
    void sync() const { check(ec_shard_group_sync(g_)); }
    // "jobs_posted", "poisoned", "blocking_issue"
    int64_t stat(const char* key) const {
        int64_t v = 0;
        check(ec_shard_group_stat(g_, key, &v));
        return v;
    }
};

class ShardedCellBuffer {
    const ShardGroup* grp_;
    CellType ct_;
    std::vector<void*> ptrs_;
    std::vector<size_t> lens_;

    ShardedCellBuffer(const ShardGroup& g, CellType ct, std::vector<size_t> lens) : grp_(&g), ct_(ct), ptrs_(lens.size(), nullptr), lens_(std::move(lens)) {
        std::vector<size_t> bytes;
        for (size_t l : lens_) bytes.push_back(l * size_of(ct));
        check(ec_sharded_alloc(grp_->raw(), bytes.data(), ptrs_.data()));
    }

public:
    ShardedCellBuffer(ShardedCellBuffer&& o) noexcept : grp_(o.grp_), ct_(o.ct_), ptrs_(std::move(o.ptrs_)), lens_(std::move(o.lens_)) { o.ptrs_.clear(); }
    ShardedCellBuffer(const ShardedCellBuffer&) = delete;
    ~ShardedCellBuffer() {
        if (!ptrs_.empty()) { ec_shard_group_sync(grp_->raw()); ec_sharded_free(grp_->raw(), ptrs_.data()); }
    }
    // `From<Vec<T>>` of a row-major n_rows x n_cols raster: contiguous row-blocks, block i to device i
    template <typename T> static ShardedCellBuffer scatter(const ShardGroup& g, const std::vector<T>& data, uint64_t n_rows, uint64_t n_cols) {
        std::vector<size_t> lens, offs, bytes;
        for (int i = 0; i < g.size(); ++i) {
            uint64_t o = 0, l = 0;
            check(ec_shard_range(n_rows, n_cols, static_cast<uint32_t>(i), static_cast<uint32_t>(g.size()), &o, &l));
            lens.push_back(l); offs.push_back(o * sizeof(T)); bytes.push_back(l * sizeof(T));
        }
        ShardedCellBuffer b(g, CellEncoding<T>::cell_type(), lens);
        check(ec_sharded_upload(g.raw(), b.ptrs_.data(), data.data(), offs.data(), bytes.data()));
        return b;
    }
    CellType cell_type() const { return ct_; }
    size_t len() const { size_t n = 0; for (size_t l : lens_) n += l; return n; }
    const std::vector<size_t>& shard_lens() const { return lens_; }
    // impl {Add,Sub,Mul,Div} for &CellBuffer on every shard (buffer.rs:324-329); operands sharded identically
    ShardedCellBuffer binop(ec_op op, const ShardedCellBuffer& rhs) const {
        if (lens_ != rhs.lens_) throw Error(EC_ERR_LENGTH, "operands must be sharded identically");
        ShardedCellBuffer out(*grp_, CellType::Float64, lens_);
        std::vector<const void*> l(ptrs_.begin(), ptrs_.end()), r(rhs.ptrs_.begin(), rhs.ptrs_.end());
        std::vector<double*> o;
        for (void* p : out.ptrs_) o.push_back(static_cast<double*>(p));
        check(ec_sharded_binop(grp_->raw(), op, static_cast<ec_dtype>(ct_), l.data(), static_cast<ec_dtype>(rhs.ct_), r.data(), lens_.data(), o.data()));
        return out;
    }
    ShardedCellBuffer operator+(const ShardedCellBuffer& r) const { return binop(EC_ADD, r); }
    ShardedCellBuffer operator-(const ShardedCellBuffer& r) const { return binop(EC_SUB, r); }
    ShardedCellBuffer operator*(const ShardedCellBuffer& r) const { return binop(EC_MUL, r); }
    ShardedCellBuffer operator/(const ShardedCellBuffer& r) const { return binop(EC_DIV, r); }
    // BufferOps::min_max of the whole raster (buffer.rs:169-173): per-shard keys, one all-reduce(MAX), decode
    std::pair<CellValue, CellValue> min_max() const {
        ec_value mn, mx;
        std::vector<const void*> p(ptrs_.begin(), ptrs_.end());
        check(ec_sharded_min_max(grp_->raw(), static_cast<ec_dtype>(ct_), p.data(), nullptr, lens_.data(), &mn, &mx));
        return {CellValue(mn), CellValue(mx)};
    }
    template <typename T> std::vector<T> to_vec() const {  // gather; T must be the buffer's own cell type
        if (CellEncoding<T>::cell_type() != ct_) throw NarrowingError(ct_, CellEncoding<T>::cell_type());
        std::vector<T> out(len());
        std::vector<size_t> offs, bytes;
        size_t acc = 0;
        for (size_t l : lens_) { offs.push_back(acc * sizeof(T)); bytes.push_back(l * sizeof(T)); acc += l; }
        std::vector<const void*> p(ptrs_.begin(), ptrs_.end());
        check(ec_sharded_download(grp_->raw(), out.data(), p.data(), offs.data(), bytes.data()));
        return out;
    }
};

}  // namespace sharded

}  // namespace erased_cells
