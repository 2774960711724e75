// test_host_mirror.cpp — the reference's unit tests, restated against the C++ host
// mirror (erased_cells.hpp) over liberased_cells_hip.so.  Each function names the
// reference test it mirrors (paths relative to the reference tree).
//
//   ./test_host_mirror --host-only   lattice / scalar tests (no GPU needed)
//   ./test_host_mirror               everything (needs an MI355X)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <limits>

#include "erased_cells.hpp"
#include "raster_io.hpp"

using namespace erased_cells;

static int g_checks = 0;
#define CHECK(cond)                                                                    \
    do {                                                                               \
        ++g_checks;                                                                    \
        if (!(cond)) {                                                                 \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)
#define CHECK_THROWS(T, expr)                                                          \
    do {                                                                               \
        ++g_checks;                                                                    \
        bool thrown_ = false;                                                          \
        try { (void)(expr); } catch (const T&) { thrown_ = true; }                     \
        if (!thrown_) {                                                                \
            std::fprintf(stderr, "FAILED %s:%d: %s did not throw %s\n", __FILE__, __LINE__, #expr, #T); \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

// a small integer as a CellValue of cell type `ct`
static CellValue of(CellType ct, int x) {
    switch (ct) {
#define OF(ID, P) case CellType::ID: return CellValue(static_cast<P>(x));
        EC_HOST_WITH_CT(OF)
#undef OF
    }
    return CellValue();
}

// ------------------------------------------------------------------ src/ctype.rs:188-278
static void ctype_tests() {
    CHECK(union_of(CellType::UInt8, CellType::UInt8) == CellType::UInt8);
    CHECK(union_of(CellType::UInt16, CellType::UInt16) == CellType::UInt16);
    CHECK(union_of(CellType::Float32, CellType::Float32) == CellType::Float32);
    CHECK(union_of(CellType::Float64, CellType::Float64) == CellType::Float64);
    CHECK(union_of(CellType::Int16, CellType::Float32) == CellType::Float32);
    CHECK(union_of(CellType::Float32, CellType::Int16) == CellType::Float32);
    CHECK(union_of(CellType::UInt8, CellType::UInt16) == CellType::UInt16);
    CHECK(union_of(CellType::Int32, CellType::Float32) == CellType::Float64);
    CHECK(is_integral(CellType::UInt8) && is_integral(CellType::UInt16));
    CHECK(!is_integral(CellType::Float32) && !is_integral(CellType::Float64));
    size_t sizes[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8};
    for (CellType ct : cell_types()) {
        CHECK(size_of(ct) == sizes[static_cast<int>(ct)]);
        CHECK(cell_type_from_str(to_string(ct)) == ct);  // can_string round trip
        CHECK(one(ct) + zero(ct) == one(ct));            // zero_one
    }
    CHECK_THROWS(ParseError, cell_type_from_str("UInt57"));
#define LIM(ID, P)                                                                   \
    CHECK(min_value(CellType::ID) == CellValue(std::numeric_limits<P>::lowest()));   \
    CHECK(max_value(CellType::ID) == CellValue(std::numeric_limits<P>::max()));
    EC_HOST_WITH_CT(LIM)  // has_min_max
#undef LIM
}

// ------------------------------------------------------------------ src/value.rs:283-391, src/encoding.rs:45-49
static void value_tests() {
#define CT(ID, P) CHECK(CellValue(P{}).cell_type() == CellType::ID);
    EC_HOST_WITH_CT(CT)  // cell_type
#undef CT
#define GET(ID, P) { P v{}; CellValue cv(v); CHECK(cv.get<P>() == v); CHECK(cv.get<double>() == static_cast<double>(v)); }
    EC_HOST_WITH_CT(GET)  // get
#undef GET
    // convert
    CHECK(CellValue(uint8_t(43)).convert(CellType::Int16).cell_type() == CellType::Int16);
    CHECK(CellValue(uint8_t(43)).convert(CellType::Int16).get<int16_t>() == 43);
    CHECK_THROWS(NarrowingError, CellValue(3.11111f).convert(CellType::Int32));
    CHECK(CellValue(3.11111f).convert(CellType::Float32).get<float>() == 3.11111f);
    CHECK(CellValue(uint16_t(33)).convert(CellType::Float32).get<float>() == 33.0f);
    try { CellValue(3.5).get<uint8_t>(); CHECK(false); } catch (const NarrowingError& e) {
        CHECK(e.src == CellType::Float64 && e.dst == CellType::UInt8);
    }
    // zero_one
    CHECK(CellValue::zero().is_zero() && !CellValue::one().is_zero());
    // unary: result types of Neg
    CHECK((-CellValue(uint8_t(1))).cell_type() == CellType::Int16 && (-CellValue(uint8_t(1))).get<int16_t>() == -1);
    CHECK((-CellValue(uint16_t(1))).cell_type() == CellType::Int32 && (-CellValue(uint16_t(1))).get<int32_t>() == -1);
    CHECK((-CellValue(int8_t(1))).cell_type() == CellType::Int8);
    CHECK((-CellValue(int16_t(1))).cell_type() == CellType::Int16);
    CHECK((-CellValue(1.0)).cell_type() == CellType::Float64 && (-CellValue(1.0f)).cell_type() == CellType::Float32);
    CHECK((-CellValue(uint32_t(7))).cell_type() == CellType::Float64 && (-CellValue(uint64_t(7))).get<double>() == -7.0);
    CHECK((-CellValue(std::numeric_limits<int8_t>::min())).get<int8_t>() == std::numeric_limits<int8_t>::min());  // release wrap
    // binops: always Float64; the reference's Float32 expectations hold through PartialEq's unify
    {
        CellValue l(uint8_t(1)), r(uint8_t(2));
        CHECK(l + r == CellValue(3.0) && (l + r).cell_type() == CellType::Float64);
        CHECK(l + CellValue(2) == CellValue(3.0) && l - r == CellValue(-1.0) && l - CellValue(2) == CellValue(-1.0));
        CHECK(r - l == CellValue(1.0) && l * r == CellValue(2.0) && r * l == CellValue(2.0));
        CHECK(l / r == CellValue(0.5) && r / l == CellValue(2.0));
    }
    {
        CellValue l(1.0f), r(2.0f);
        CHECK(l + r == CellValue(3.0f) && l - r == CellValue(-1.0f) && l * r == CellValue(2.0f) && l / r == CellValue(0.5f));
        CHECK((l / r).cell_type() == CellType::Float64);
    }
    // total order: NaN == NaN bitwise, -0 < +0
    CHECK(CellValue(std::nan("")) == CellValue(std::nan("")));
    CHECK(CellValue(-0.0) < CellValue(0.0));
    CHECK(CellValue(uint64_t(1) << 63) > CellValue(uint64_t(5)));
}

// ------------------------------------------------------------------ src/masked/nodata.rs:75-95
static void nodata_tests() {
    CHECK(!NoData<int16_t>::None().value().has_value());
    CHECK(NoData<uint8_t>::Default().value() == uint8_t(0));
    CHECK(std::isnan(*NoData<float>::Default().value()));
    CHECK(NoData<uint16_t>::new_(6).value() == uint16_t(6));
#define DEF(ID, P) CHECK(NoData<P>::Default().value().has_value());
    EC_HOST_WITH_CT(DEF)
#undef DEF
    CHECK(NoData<double>::Default().is(CellValue(std::numeric_limits<double>::quiet_NaN())));
    CHECK(NoData<int16_t>::Default().value() == std::numeric_limits<int16_t>::min());
}

// ------------------------------------------------------------------ examples/quick.rs, examples/buffer.rs
static void examples() {
    {
        CellBuffer buf1 = CellBuffer::from_vec<uint8_t>({1, 2, 3});
        CellBuffer buf2 = CellBuffer::from_vec<uint16_t>({2, 4, 6});
        CellBuffer result = buf1 / buf2 * 0.5;
        CHECK(result == CellBuffer::from_vec<double>({0.25, 0.25, 0.25}));
    }
    {
        CellBuffer buf1 = CellBuffer::fill_via<uint8_t>(9, [](size_t i) { return uint8_t(i); });
        CHECK(buf1.cell_type() == CellType::UInt8);
        CHECK(buf1.get(3) == CellValue(uint8_t(3)) && buf1.get(3).cell_type() == CellType::UInt8);
        auto [mn, mx] = buf1.min_max();
        CHECK(mn == CellValue(uint8_t(0)) && mx == CellValue(uint8_t(8)) && mn.cell_type() == CellType::UInt8);
        CHECK(((mx - mn + CellValue(1)) / CellValue(2)) == CellValue(4.5));
        CellBuffer buf2 = CellBuffer::fill_via<float>(9, [](size_t i) { return 8.0f - float(i); });
        CHECK(buf2.cell_type() == CellType::Float32);
        auto mm2 = buf2.min_max();
        CHECK(mm2.first == CellValue(0.0f) && mm2.second == CellValue(8.0f) && mm2.first.cell_type() == CellType::Float32);
        CellBuffer diff = buf2 - buf1;
        auto mm3 = diff.min_max();
        CHECK(mm3.first == CellValue(-8) && mm3.second == CellValue(8));
    }
}

// ------------------------------------------------------------------ src/buffer.rs:461-672
static void buffer_tests() {
    for (CellType ct : cell_types()) {  // defaults, put_get
        CellBuffer cv = CellBuffer::with_defaults(3, ct);
        CHECK(cv.len() == 3 && cv.get(0) == zero(ct) && cv.cell_type() == ct);
        CellBuffer b = CellBuffer::fill(3, zero(ct));
        b.put(1, one(ct));
        CHECK(b.get(1) == one(ct).convert(ct));
    }
    CHECK_THROWS(NarrowingError, CellBuffer::with_defaults(3, CellType::UInt8).put(0, CellValue(1.5f)));
    CHECK_THROWS(std::out_of_range, CellBuffer::with_defaults(3, CellType::UInt8).get(3));
#define TOVEC(ID, P) { std::vector<P> v(3, P{}); CHECK(CellBuffer::from_vec(v).to_vec<P>() == v); }
    EC_HOST_WITH_CT(TOVEC)  // to_vec
#undef TOVEC
    {  // min_max
        auto [mn, mx] = CellBuffer::from_vec<double>({-1.0, 3.0, 2000.0, -5555.5}).min_max();
        CHECK(mn == CellValue(-5555.5) && mx == CellValue(2000.0) && mn.cell_type() == CellType::Float64);
        auto mm = CellBuffer::from_vec<uint8_t>({1, 3, 200, 0}).min_max();
        CHECK(mm.first == CellValue(uint8_t(0)) && mm.second == CellValue(uint8_t(200)));
    }
    for (CellType ct : cell_types()) {  // convert
        CellBuffer buf = CellBuffer::with_defaults(3, ct);
        for (CellType target : cell_types()) {
            if (can_fit_into(ct, target)) CHECK(buf.convert(target).cell_type() == target);
            else CHECK_THROWS(NarrowingError, buf.convert(target));
        }
    }
    for (CellType ct : cell_types()) {  // unary
        CellBuffer buf = -CellBuffer::fill(3, one(ct));
        CHECK(buf.get(0) == -one(ct) && buf.cell_type() == (-one(ct)).cell_type());
    }
    for (CellType lct : cell_types())  // binary: all 100 pairs, 4 ops, both orders
        for (CellType rct : cell_types()) {
            CellValue lv = one(lct), rv = one(rct) + one(rct);
            CellBuffer lhs = CellBuffer::fill(3, lv), rhs = CellBuffer::fill(3, rv);
            CHECK((lhs + rhs).get(0) == lv + rv && (rhs + lhs).get(1) == rv + lv);
            CHECK((lhs - rhs).get(2) == lv - rv && (rhs - lhs).get(0) == rv - lv);
            CHECK((lhs * rhs).get(1) == lv * rv && (rhs * lhs).get(2) == rv * lv);
            CHECK((lhs / rhs).get(0) == lv / rv && (rhs / lhs).get(1) == rv / lv);
            CellBuffer typed = CellBuffer::fill(3, of(rct, 2));
            CHECK((lhs / typed).get(2) == CellValue(0.5) && (lhs / typed).cell_type() == CellType::Float64);
        }
    {  // scalar
        CellBuffer buf = CellBuffer::fill_via<uint8_t>(9, [](size_t i) { return uint8_t(i + 1); });
        CHECK(buf * 2.0 == CellBuffer::fill_via<double>(9, [](size_t i) { return (double(i) + 1.0) * 2.0; }));
    }
    {  // equal, cmp
        CellBuffer buf = CellBuffer::fill_via<double>(9, [](size_t i) { return i % 2 == 0 ? std::nan("") : double(i); });
        CHECK(buf == buf);
        auto wd = CellBuffer::with_defaults;
        CHECK(wd(4, CellType::UInt8) == wd(4, CellType::UInt8) && wd(4, CellType::UInt8) != wd(5, CellType::UInt8));
        CHECK(CellBuffer::from_vec<int32_t>({1, 2, 3}) < CellBuffer::from_vec<int32_t>({2, 3, 4}));
        CHECK(CellBuffer::from_vec<int32_t>({1, 2, 3}) < CellBuffer::from_vec<int32_t>({2, 3}));
        CHECK(CellBuffer::from_vec<double>({std::nan(""), 2.0, 3.0}) < CellBuffer::from_vec<double>({std::nan(""), 2.0, 4.0}));
        CHECK(wd(4, CellType::UInt8) < wd(4, CellType::Float32) && wd(4, CellType::Float32) > wd(4, CellType::UInt8));
        CHECK(wd(4, CellType::UInt8) < wd(5, CellType::UInt8) && wd(5, CellType::Float64) > wd(4, CellType::Float64));
    }
    {  // from_others (buffer.rs:528-555)
        CellBuffer b = CellBuffer::from_values({CellValue(uint16_t(3)), CellValue(uint16_t(4)), CellValue(uint16_t(5))});
        CHECK(b.cell_type() == CellType::UInt16 && b.len() == 3 && b.get(2) == CellValue(uint16_t(5)));
        b = CellBuffer::from_vec(std::vector<float>{33.3f, 44.4f, 55.5f});
        CHECK(b.cell_type() == CellType::Float32 && b.len() == 3 && b.get(2) == CellValue(55.5f));
        CHECK(CellBuffer::from_values({}).cell_type() == CellType::UInt8);
        CHECK(CellBuffer::from_values({CellValue(int32_t(7)), CellValue(uint8_t(9))}).to_vec<int32_t>() == (std::vector<int32_t>{7, 9}));
        CHECK_THROWS(NarrowingError, CellBuffer::from_values({CellValue(uint8_t(7)), CellValue(int32_t(9))}));
        auto vals = b.to_values();
        CHECK(vals.size() == 3 && vals[0] == CellValue(33.3f) && vals[2] == CellValue(55.5f));
    }
    {  // debug (buffer.rs:558-564), elided (lib.rs:197-206)
        CHECK(elided(std::vector<std::string>(3, "1")) == "1, 1, 1");
        CHECK(elided(std::vector<std::string>(30, "0")) == "0, 0, 0, 0, 0, ... 0, 0, 0, 0, 0");
        CellBuffer b = CellBuffer::fill(5, CellValue(37));
        CHECK(b.debug_string() == "Int32CellBuffer(37, 37, 37, 37, 37)");
        b = CellBuffer::fill(15, CellValue(37));
        CHECK(b.debug_string() == "Int32CellBuffer(37, 37, 37, 37, 37, ... 37, 37, 37, 37, 37)");
        CHECK(CellBuffer::from_vec(std::vector<float>{0.5f, 2.0f}).debug_string() == "Float32CellBuffer(0.5, 2.0)");
    }
    {  // extend (buffer.rs:489-498)
        CellBuffer buf = CellBuffer::fill(3, CellValue(uint8_t(0)));
        CHECK(!buf.is_empty() && buf.cell_type() == CellType::UInt8);
        buf.extend(std::vector<int>{1});
        CHECK(buf.cell_type() == CellType::UInt8 && buf.len() == 4 && buf.get(0) == CellValue(0) && buf.get(3) == CellValue(1));
        CHECK_THROWS(std::overflow_error, buf.extend(std::vector<int>{300}));
    }
    {  // shape rules: zip truncation, empty results are UInt8 (buffer.rs:327, :233-234)
        CellBuffer a = CellBuffer::with_defaults(5, CellType::Int16), b = CellBuffer::with_defaults(3, CellType::Float32);
        CHECK((a + b).len() == 3);
        CellBuffer e = CellBuffer::with_defaults(0, CellType::UInt16);
        CHECK((e + a).cell_type() == CellType::UInt8 && (e + a).len() == 0 && (-e).cell_type() == CellType::UInt8);
        CHECK(e.convert(CellType::Float32).cell_type() == CellType::UInt8);
        CHECK_THROWS(std::logic_error, e.to_vec<float>());  // danger::cast assert in the reference
    }
}

// ------------------------------------------------------------------ src/masked/mask.rs:184-242
static void mask_tests() {
    CHECK(Mask::fill(3, true).counts() == std::make_pair(size_t(3), size_t(0)));
    CHECK(Mask::fill(3, false).counts() == std::make_pair(size_t(0), size_t(3)));
    CHECK(Mask::fill_via(3, [](size_t i) { return i % 2 == 0; }).counts() == std::make_pair(size_t(2), size_t(1)));
    Mask m = Mask::fill(3, true);
    m.put(1, false);
    m.put(0, false);
    CHECK(m == Mask::new_({false, false, true}));
    Mask t = Mask::fill(4, true), f = Mask::fill(4, false);
    CHECK((!t) == f);
    CHECK((!Mask::fill(4, true)) == f);  // consuming form
    Mask mm = Mask::new_({true, false, true, false}), rr = Mask::new_({false, true, false, true});
    CHECK((!mm) == rr);
    Mask alt = Mask::fill_via(4, [](size_t i) { return i % 2 == 0; });
    CHECK(!alt.all(true) && !alt.all(false) && t.all(true) && !t.all(false));
    Mask l = Mask::fill_via(4, [](size_t i) { return i % 2 == 0; }), r = Mask::fill_via(4, [](size_t i) { return i % 2 != 0; });
    CHECK((l & r).all(false) && (l | r).all(true));
    CHECK((l.clone() & r).all(false) && (l.clone() | r).all(true));  // consuming forms
    Mask m5 = Mask::fill(5, true), m3 = Mask::fill(3, false);
    CHECK((m5 & m3).len() == 3 && (m5 | m3).len() == 3);  // &Mask & &Mask zips to the shorter (mask.rs:129-140)
    Mask owned = Mask::fill(5, true) & m3;                // Mask & Mask mutates lhs in place, length kept (mask.rs:118-127)
    CHECK(owned == Mask::new_({false, false, false, true, true}));
}

// ------------------------------------------------------------------ src/masked/masked_buffer.rs:400-541, examples/masked.rs
static std::pair<uint8_t, bool> filler_masker(size_t i) { return {uint8_t(i), i % 2 == 0}; }

static void masked_tests() {
    {  // ctor
        MaskedCellBuffer m = MaskedCellBuffer::fill_via<uint8_t>(3, [](size_t i) { return uint8_t(i); });
        MaskedCellBuffer r = MaskedCellBuffer::new_(CellBuffer::fill_via<uint8_t>(3, [](size_t i) { return uint8_t(i); }), Mask::fill(3, true));
        CHECK(m == r);
        CHECK(MaskedCellBuffer::from_vec<double>({0, 0, 0, 0}).mask().counts().first == 4);
        CHECK(MaskedCellBuffer::with_defaults(4, CellType::Int16).mask().counts().first == 4);
        CHECK_THROWS(std::logic_error, MaskedCellBuffer::new_(CellBuffer::with_defaults(4, CellType::UInt8), Mask::fill(3, true)));
    }
    {  // vec_with_nodata
        double nan = std::numeric_limits<double>::quiet_NaN();
        std::vector<double> v = {1.0, nan, 3.0, nan};
        MaskedCellBuffer m = MaskedCellBuffer::from_vec_with_nodata(v, NoData<double>::Default());
        CHECK(m == MaskedCellBuffer::new_(CellBuffer::from_vec(v), Mask::new_({true, false, true, false})));
        MaskedCellBuffer m2 = MaskedCellBuffer::from_vec_with_nodata(v, NoData<double>::new_(3.0));
        CHECK(m2 == MaskedCellBuffer::new_(CellBuffer::from_vec(v), Mask::new_({true, true, false, true})));
        CHECK(MaskedCellBuffer::from_vec_with_nodata(v, NoData<double>::None()).mask().all(true));
    }
    {  // get_masked
        MaskedCellBuffer buf = MaskedCellBuffer::fill_with_mask_via<uint8_t>(9, filler_masker);
        CHECK(buf.get(4) == CellValue(4) && buf.get_masked(4) == CellValue(4) && !buf.get_masked(5).has_value());
        buf.put(5, CellValue(uint8_t(4)));
        CHECK(!buf.get_masked(5).has_value());
        buf.mask_mut().put(5, true);
        CHECK(buf.get_masked(5) == CellValue(4));
        buf.put_with_mask(5, CellValue(uint8_t(99)), false);
        CHECK(!buf.get_masked(5).has_value());
    }
    {  // debug (masked_buffer.rs:533-540)
        MaskedCellBuffer m = MaskedCellBuffer::from_vec(std::vector<int32_t>{0});
        CHECK(m.debug_string() == "Int32MaskedCellBuffer(Int32CellBuffer(0), Mask(true))");
    }
    {  // derived PartialOrd over (buffer, mask) (masked_buffer.rs:39)
        MaskedCellBuffer a = MaskedCellBuffer::new_(CellBuffer::from_vec(std::vector<uint8_t>{1, 2, 3}), Mask::new_({true, false, true}));
        MaskedCellBuffer b = MaskedCellBuffer::new_(CellBuffer::from_vec(std::vector<uint8_t>{1, 2, 3}), Mask::new_({true, true, true}));
        MaskedCellBuffer c = MaskedCellBuffer::new_(CellBuffer::from_vec(std::vector<uint8_t>{1, 2, 4}), Mask::new_({false, false, false}));
        CHECK(a.cmp(a) == 0 && a < b && b > a && b < c && a < c && !(c < a));
    }
    {  // extend (masked_buffer.rs:449-455)
        MaskedCellBuffer buf = MaskedCellBuffer::fill(3, CellValue(0));
        buf.extend(std::vector<std::pair<int, bool>>{{1, false}});
        CHECK(buf.get_masked(0) == CellValue(0) && !buf.get_masked(3).has_value() && buf.len() == 4);
    }
    {  // convert
        MaskedCellBuffer buf = MaskedCellBuffer::fill_with_mask_via<uint8_t>(4, filler_masker);
        CHECK(buf.convert(CellType::Float64).to_vec<double>() == std::vector<double>({0.0, 1.0, 2.0, 3.0}));
    }
    {  // unary
        MaskedCellBuffer mbuf = MaskedCellBuffer::fill_with_mask_via<uint8_t>(9, filler_masker);
        int16_t mn = std::numeric_limits<int16_t>::min();
        CHECK((-mbuf).to_vec_with_nodata(NoData<int16_t>::Default()) == std::vector<int16_t>({0, mn, -2, mn, -4, mn, -6, mn, -8}));
    }
    {  // min_max
        MaskedCellBuffer mbuf = MaskedCellBuffer::fill_with_mask_via<uint8_t>(9, [](size_t i) { return std::make_pair(uint8_t(i), i != 0 && i != 8); });
        auto [mn, mx] = mbuf.min_max();
        CHECK(mn == CellValue(uint8_t(1)) && mx == CellValue(uint8_t(7)) && mn.cell_type() == CellType::UInt8);
    }
    {  // scalar
        MaskedCellBuffer all = MaskedCellBuffer::fill_with_mask_via<uint8_t>(9, [](size_t i) { return std::make_pair(uint8_t(i), true); });
        CellBuffer expected = CellBuffer::fill_via<uint8_t>(9, [](size_t i) { return uint8_t(i); }) * 2.0;
        CHECK(all * 2.0 == MaskedCellBuffer::from(expected.clone()));
        MaskedCellBuffer r = MaskedCellBuffer::fill_with_mask_via<uint8_t>(9, filler_masker) * 2.0;
        CHECK(r != MaskedCellBuffer::from(expected.clone()));
        double fmin = std::numeric_limits<double>::lowest();
        CHECK(r.to_vec_with_nodata(NoData<double>::new_(fmin)) == std::vector<double>({0.0, fmin, 4.0, fmin, 8.0, fmin, 12.0, fmin, 16.0}));
    }
    {  // binary
        MaskedCellBuffer lhs = MaskedCellBuffer::new_(CellBuffer::fill(9, CellValue(1.0)), Mask::fill_via(9, [](size_t i) { return i % 2 == 0; }));
        MaskedCellBuffer rhs = MaskedCellBuffer::new_(CellBuffer::fill(9, CellValue(2.0)), Mask::fill(9, true));
        CHECK((lhs + rhs).get_masked(0) == CellValue(3.0) && !(lhs + rhs).get_masked(1).has_value());
        CHECK((lhs - rhs).get_masked(2) == CellValue(-1.0) && !(lhs - rhs).get_masked(3).has_value());
        CHECK((lhs * rhs).get_masked(4) == CellValue(2.0) && !(lhs * rhs).get_masked(5).has_value());
        CHECK((lhs / rhs).get_masked(6) == CellValue(0.5) && !(lhs / rhs).get_masked(7).has_value());
    }
    {  // examples/masked.rs
        MaskedCellBuffer buf = MaskedCellBuffer::fill_with_mask_via<double>(4, [](size_t i) { return std::make_pair(double(i), i % 2 == 0); });
        CHECK(buf.mask() == Mask::new_({true, false, true, false}));
        CHECK(buf.counts() == std::make_pair(size_t(2), size_t(2)));
        MaskedCellBuffer ones = MaskedCellBuffer::from_vec<double>({1.0, 1.0, 1.0, 1.0});
        MaskedCellBuffer r = (buf + ones) * 2.0;
        CHECK(r == MaskedCellBuffer::new_(CellBuffer::from_vec<double>({2.0, 4.0, 6.0, 8.0}), Mask::new_({true, false, true, false})));
    }
}

// ------------------------------------------------------------------ src/gdal/rasterband.rs:20-36, 57-71, 138-191; src/gdal/mod.rs:49-70
static void gdal_tests(const std::string& data_dir) {
    auto path = [&](const char* name) { return data_dir + "/" + name; };
    {  // doc-test read_cells: buffer.min_max() equals the band's min/max
        RasterBand rb = RasterBand::open(path("L8-Elkton-VA-B5.tiff"));
        CHECK(rb.size() == std::make_pair(size_t(186), size_t(169)) && rb.band_type() == CellType::UInt16);
        auto [mn, mx] = rb.read_cells().min_max();
        CHECK(mn == CellValue(uint16_t(5469)) && mx == CellValue(uint16_t(39368)) && mn.cell_type() == CellType::UInt16);
    }
    {  // read_cells: NDVI against the gdal_calc numbers quoted in the reference
        CellBuffer red = RasterBand::open(path("L8-Elkton-VA-B4.tiff")).read_cells();
        CellBuffer nir = RasterBand::open(path("L8-Elkton-VA-B5.tiff")).read_cells();
        CellBuffer ndvi = (nir - red) / (nir + red);
        auto [mn, mx] = ndvi.min_max();
        CHECK(mn.to_f64() - -0.1248899911993 < 1e-8 && std::fabs(mn.to_f64() - -0.1248899911993) < 1e-8);
        CHECK(mx.to_f64() - 0.66998345719859 < 1e-8 && std::fabs(mx.to_f64() - 0.66998345719859) < 1e-8);
        CHECK(mn.to_f64() == -0x1.ff8ca5bcc77dcp-4 && mx.to_f64() == 0x1.5708125b0ed28p-1);
    }
    {  // read_cells_masked: the NIR band has 4 nodata cells, as has the result
        RasterBand nir_rb = RasterBand::open(path("L8-Elkton-VA-B5-nd.tiff"));
        CHECK(nir_rb.no_data_value().has_value() && *nir_rb.no_data_value() == 0.0);
        MaskedCellBuffer red = RasterBand::open(path("L8-Elkton-VA-B4.tiff")).read_cells_masked();
        MaskedCellBuffer nir = nir_rb.read_cells_masked();
        auto [nir_data, nir_nodata] = nir.counts();
        CHECK(nir_data + nir_nodata == 186 * 169 && nir_nodata == 4);
        MaskedCellBuffer ndvi = (nir - red) / (nir + red);
        CHECK(ndvi.counts() == std::make_pair(nir_data, nir_nodata));
        auto [mn, mx] = ndvi.min_max();
        CHECK(mn.to_f64() == -0x1.ff8ca5bcc77dcp-4 && mx.to_f64() == 0x1.5708125b0ed28p-1);
        // row-block ingest: 8 shards (22,21,...,21 rows) tile the band
        size_t rows = 169, row0 = 0, total_nodata = 0;
        for (size_t g = 0; g < 8; ++g) {
            uint64_t off = 0, len = 0;
            check(ec_shard_range(rows, 186, static_cast<uint32_t>(g), 8, &off, &len));
            CHECK(off == row0 * 186);
            MaskedCellBuffer part = nir_rb.read_cells_masked_rows(row0, len / 186);
            total_nodata += part.counts().second;
            row0 += len / 186;
        }
        CHECK(row0 == rows && total_nodata == 4);
    }
    {  // fused single-pass chains == the eager chains, bit for bit
        using fused::lazy;
        CellBuffer red = RasterBand::open(path("L8-Elkton-VA-B4.tiff")).read_cells();
        CellBuffer nir = RasterBand::open(path("L8-Elkton-VA-B5.tiff")).read_cells();
        CellBuffer eager = (nir - red) / (nir + red);
        CHECK(fused::ndvi(nir, red) == eager);
        CHECK(fused::eval((lazy(nir) - lazy(red)) / (lazy(nir) + lazy(red))) == eager);
        CellBuffer r32 = red.convert(CellType::Float32);
        CHECK(fused::eval((lazy(nir) + lazy(r32)) * lazy(red)) == (nir + r32) * red);
        CHECK(fused::eval(lazy(nir) * lazy(r32)) == nir * r32);
        CHECK(fused::eval((lazy(nir) + lazy(red)) * 2.0) == (nir + red) * 2.0);  // examples/masked.rs:12 shape
        MaskedCellBuffer mred = RasterBand::open(path("L8-Elkton-VA-B4.tiff")).read_cells_masked();
        MaskedCellBuffer mnir = RasterBand::open(path("L8-Elkton-VA-B5-nd.tiff")).read_cells_masked();
        MaskedCellBuffer mf = fused::eval((lazy(mnir) - lazy(mred)) / (lazy(mnir) + lazy(mred)));
        CHECK(mf == (mnir - mred) / (mnir + mred) && mf.counts().second == 4);
        CHECK(fused::eval((lazy(mnir) + lazy(mred)) * 2.0) == (mnir + mred) * 2.0);
        // a tree deeper than two levels as ONE pass (ec_expr): EVI with the red band standing in for blue
        using namespace fused;
        const std::vector<ec_expr_step> evi = {{EC_SUB, stream(0), stream(1), 0}, {EC_MUL, reg(0), scalar(0), 0},
                                               {EC_MUL, stream(1), scalar(1), 1}, {EC_ADD, stream(0), reg(1), 1},
                                               {EC_MUL, stream(2), scalar(2), 2}, {EC_SUB, reg(1), reg(2), 1},
                                               {EC_ADD, reg(1), scalar(3), 1},    {EC_DIV, reg(0), reg(1), 0}};
        const std::vector<CellValue> k = {2.5, 6.0, 7.5, 1.0};
        CellBuffer evi_eager = ((nir - red) * 2.5) / (((nir + red * 6.0) - r32 * 7.5) + 1.0);
        CHECK(program(std::vector<const CellBuffer*>{&nir, &red, &r32}, k, evi) == evi_eager);
        MaskedCellBuffer mr32 = mred.convert(CellType::Float32);
        MaskedCellBuffer mevi = program(std::vector<const MaskedCellBuffer*>{&mnir, &mred, &mr32}, k, evi);
        CHECK(mevi == ((mnir - mred) * 2.5) / (((mnir + mred * 6.0) - mr32 * 7.5) + 1.0));
        CHECK_THROWS(Error, program(std::vector<const CellBuffer*>{&nir}, k, {{EC_ADD, stream(0), reg(0), 0}}));  // register read before written
        {   // host memory in, host memory out: the same program streamed through the GPU equals the resident result
            const std::vector<uint16_t> hn = nir.to_vec<uint16_t>(), hr = red.to_vec<uint16_t>();
            const std::vector<float> hf = r32.to_vec<float>();
            const std::vector<double> streamed = program_host({{CellType::UInt16, hn.data()}, {CellType::UInt16, hr.data()}, {CellType::Float32, hf.data()}},
                                                              hn.size(), k, evi, 4099);
            CHECK(CellBuffer::from_vec(streamed) == evi_eager);
        }
        CHECK(program_min_max(std::vector<const CellBuffer*>{&nir, &red, &r32}, k, evi) == evi_eager.min_max());       // statistics without the raster
        CHECK(program_min_max(std::vector<const MaskedCellBuffer*>{&mnir, &mred, &mr32}, k, evi) == mevi.min_max());
        CHECK(((tree(nir) - red) / (tree(nir) + red)).min_max() == ((nir - red) / (nir + red)).min_max());  // NDVI's statistics, no raster
        // the same tree in operator syntax: scheduled onto the program's registers, one launch
        CHECK(((tree(nir) - red) * 2.5 / (tree(nir) + tree(red) * 6.0 - tree(r32) * 7.5 + 1.0)).eval() == evi_eager);
        CHECK((((tree(mnir) - mred) * 2.5) / (((tree(mnir) + tree(mred) * 6.0) - tree(mr32) * 7.5) + 1.0)).eval() == mevi);
        CHECK((1.0 - tree(nir) / (tree(red) + 0.5)).eval() == fused::program(std::vector<const CellBuffer*>{&nir, &red}, {1.0, 0.5},
              {{EC_ADD, stream(1), scalar(1), 0}, {EC_DIV, stream(0), reg(0), 0}, {EC_SUB, scalar(0), reg(0), 0}}));  // scalar on the left
        {   // five distinct buffers: more than the four streams of a program -> cut into sub-trees, same cells
            CellBuffer r64 = red.convert(CellType::Float64), n32 = nir.convert(CellType::Float32), n64 = nir.convert(CellType::Float64);
            CellBuffer wide = ((tree(nir) + red) * (tree(r32) - r64) / ((tree(n32) + 2.0) * 3.0 + n64)).eval();
            CHECK(wide == ((nir + red) * (r32 - r64)) / (((n32 + 2.0) * 3.0) + n64));
        }
    }
    // GdalND -> NoData<T> (src/gdal/mod.rs:49-70): range-checked
    CHECK(!nodata_from_f64<uint16_t>(std::nullopt, "u16").value().has_value());
    CHECK(nodata_from_f64<uint16_t>(0.0, "u16").value() == uint16_t(0));
    CHECK(nodata_from_f64<uint8_t>(255.9, "u8").value() == uint8_t(255));
    CHECK(nodata_from_f64<float>(-9999.0, "f32").value() == -9999.0f);
    CHECK_THROWS(NoDataConversionError, nodata_from_f64<uint8_t>(-9999.0, "u8"));
    CHECK_THROWS(NoDataConversionError, nodata_from_f64<int16_t>(std::nan(""), "i16"));
    CHECK_THROWS(Error, RasterBand::open(path("does-not-exist.tiff")));
}

// examples/macros.rs: the `with_ct!` type table stamps out a match over every cell type
static const char* primitive_name(CellType ct) {
    switch (ct) {
#define EC_NAME(ID, P) case CellType::ID: return #P;
        EC_HOST_WITH_CT(EC_NAME)
#undef EC_NAME
    }
    return "?";
}

static void debug_format_checks() {
    CHECK(std::string(primitive_name(CellType::Float32)) == "float" && std::string(primitive_name(CellType::UInt16)) == "uint16_t");  // Rust `{:?}` of floats; no device needed
    CHECK(rust_debug(0.25) == "0.25" && rust_debug(37.0) == "37.0" && rust_debug(-0.0) == "-0.0");
    CHECK(rust_debug(1e-7) == "1e-7" && rust_debug(1e16) == "1e16" && rust_debug(1.5e300) == "1.5e300");
    CHECK(rust_debug(123456789012345680.0) == "1.2345678901234568e17" && rust_debug(0.0001) == "0.0001" && rust_debug(0.00001) == "1e-5");
    CHECK(rust_debug(0.1f) == "0.1" && rust_debug(16777216.0f) == "16777216.0" && rust_debug(1e-10f) == "1e-10");
    CHECK(rust_debug(std::nan("")) == "NaN" && rust_debug(INFINITY) == "inf" && rust_debug(-HUGE_VAL) == "-inf");
    CHECK(rust_debug<int8_t>(-5) == "-5" && rust_debug<uint8_t>(200) == "200" && rust_debug<uint64_t>(18446744073709551615ull) == "18446744073709551615");
}

// One process driving "several" GPUs: a 1-GPU box lists device 0 three times under EC_GROUP_HOST_COMBINE (launch
// threads, row-block scatter/gather, fan-out and the combine all run; only the xGMI hop does not), and once over RCCL.
static void sharded_tests() {
    using namespace sharded;
    const uint64_t rows = 37, cols = 53;
    std::vector<uint16_t> x(rows * cols), d(rows * cols);
    for (size_t i = 0; i < x.size(); ++i) { x[i] = static_cast<uint16_t>(1000 + (i * 7919u) % 50000u); d[i] = static_cast<uint16_t>(1 + (i * 104729u) % 65535u); }
    x[5] = 3; x[x.size() - 9] = 65000;  // extremes in the first and the last shard
    CellBuffer whole_x = CellBuffer::from_vec(x), whole_d = CellBuffer::from_vec(d);
    const auto want_q = (whole_x / whole_d).to_vec<double>();
    const auto want_mm = whole_x.min_max();
    for (int variant = 0; variant < 2; ++variant) {
        ShardGroup g(variant == 0 ? std::vector<int32_t>{0, 0, 0} : std::vector<int32_t>{0},
                     variant == 0 ? EC_GROUP_HOST_COMBINE : EC_GROUP_RCCL);
        auto sx = ShardedCellBuffer::scatter(g, x, rows, cols), sd = ShardedCellBuffer::scatter(g, d, rows, cols);
        CHECK(sx.len() == x.size() && sx.cell_type() == CellType::UInt16);
        if (variant == 0) CHECK((sx.shard_lens() == std::vector<size_t>{13 * cols, 12 * cols, 12 * cols}));
        auto q = sx / sd;
        CHECK(q.cell_type() == CellType::Float64);
        const auto got = q.to_vec<double>();
        CHECK(got.size() == want_q.size() && std::memcmp(got.data(), want_q.data(), got.size() * sizeof(double)) == 0);
        const auto mm = sx.min_max();
        CHECK(mm.first == want_mm.first && mm.second == want_mm.second);
        CHECK(mm.first == CellValue(uint16_t(3)) && mm.second == CellValue(uint16_t(65000)));
        CHECK(sx.to_vec<uint16_t>() == x);
    }
}

int main(int argc, char** argv) {
    bool host_only = argc > 1 && std::string(argv[1]) == "--host-only";
    const char* dd = std::getenv("TEST_DATA_DIR");  // as the reference's testkit (.cargo/config.toml:3)
    try {
        ctype_tests();
        value_tests();
        debug_format_checks();
        if (!host_only) {
            init(0);
            nodata_tests();
            examples();
            buffer_tests();
            mask_tests();
            masked_tests();
            sharded_tests();
            if (dd) gdal_tests(dd);
        } else {
            nodata_tests();
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "FAILED with exception: %s\n", e.what());
        return 1;
    }
    std::printf("host mirror: %d checks passed%s%s\n", g_checks, host_only ? " (host-only subset)" : "",
                (!host_only && dd) ? " (incl. GDAL fixture tests)" : "");
    return 0;
}
