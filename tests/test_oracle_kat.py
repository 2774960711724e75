"""Pin the oracle against every known-answer the reference's own tests, doc-tests
and examples hold for the hot path.

Each test cites the reference assertion it restates (paths relative to
/root/reference).  No GPU; the oracle is the thing under test here.
"""
import math
import os

import numpy as np
import pytest

from oracle import eco
from oracle.eco import (ADD, DIV, F32, F64, I8, I16, I32, I64, MUL, ND_DEFAULT, ND_NONE,
                        ND_VALUE, NP_DTYPES, NTYPES, SUB, U8, U16, U32, U64, Value)
from tiff_util import read_tiff

OPS = [ADD, SUB, MUL, DIV]
PYOP = {ADD: lambda a, b: a + b, SUB: lambda a, b: a - b, MUL: lambda a, b: a * b, DIV: lambda a, b: a / b}


def V(ct, x):
    return Value.of(ct, x)


def one(ct):
    return V(ct, 1)


# ---- B.1 examples/quick.rs:5-11, README.md:19-32 ---------------------------
def test_b1_quick_example():
    r = eco.binop(DIV, np.array([1, 2, 3], np.uint8), np.array([2, 4, 6], np.uint16))
    assert r.dtype == np.float64
    r2 = eco.binop_scalar(MUL, r, V(F64, 0.5))
    assert r2.dtype == np.float64 and r2.tolist() == [0.25, 0.25, 0.25]


# ---- B.2 / B.3 examples/buffer.rs:5-30, src/buffer.rs:23-48 -----------------
def test_b2_b3_buffer_example():
    buf1 = np.arange(9, dtype=np.uint8)
    mn, mx = eco.min_max(buf1)
    assert (mn.ct, mn.get(), mx.ct, mx.get()) == (U8, 0, U8, 8)
    # ((max - min + 1) / 2) == 4.5   (i32 scalars: `1`, `2`)
    t = eco.value_binop(SUB, mx, mn)
    t = eco.value_binop(ADD, t, V(I32, 1))
    t = eco.value_binop(DIV, t, V(I32, 2))
    assert t.ct == F64 and t.get() == 4.5
    buf2 = (8 - np.arange(9)).astype(np.float32)
    mn, mx = eco.min_max(buf2)
    assert (mn.ct, mn.get(), mx.ct, mx.get()) == (F32, 0.0, F32, 8.0)
    diff = eco.binop(SUB, buf2, buf1)
    mn, mx = eco.min_max(diff)
    assert eco.value_eq(mn, V(I32, -8)) and eco.value_eq(mx, V(I32, 8))
    assert mn.ct == F64


# ---- B.4 src/value.rs:349-391 ------------------------------------------------
@pytest.mark.parametrize("ct", [U8, U16, F32, F64])
def test_b4_value_binops(ct):
    l, r = V(ct, 1), V(ct, 2)
    exp = {(ADD, 0): 3.0, (SUB, 0): -1.0, (SUB, 1): 1.0, (MUL, 0): 2.0, (MUL, 1): 2.0, (DIV, 0): 0.5, (DIV, 1): 2.0}
    for (op, swap), e in exp.items():
        a, b = (r, l) if swap else (l, r)
        res = eco.value_binop(op, a, b)
        assert res.ct == F64, "every binop yields Float64 (value.rs:207, :24-33)"
        assert res.get() == e
        # the reference asserts against CellValue::Float32(e) for f32 operands; PartialEq unifies
        assert eco.value_eq(res, V(ct if ct in (F32, F64) else F64, e))
    # `l + 2`, `l - 2` with an i32 scalar (value.rs:353,355)
    assert eco.value_binop(ADD, V(U8, 1), V(I32, 2)).get() == 3.0
    assert eco.value_binop(SUB, V(U8, 1), V(I32, 2)).get() == -1.0


# ---- B.5 src/value.rs:338-346 ------------------------------------------------
def test_b5_neg_result_types():
    cases = [(U8, I16), (U16, I32), (I8, I8), (I16, I16), (F64, F64), (F32, F32)]
    for src, dst in cases:
        r = eco.value_neg(V(src, 1))
        assert r.ct == dst and r.get() == -1
    assert eco.value_neg(V(U32, 7)).ct == F64 and eco.value_neg(V(U64, 7)).get() == -7.0
    assert eco.value_neg(V(I32, 5)).ct == I32 and eco.value_neg(V(I64, 5)).get() == -5


# ---- B.6 src/value.rs:313-329 ------------------------------------------------
def test_b6_value_convert():
    r = eco.value_convert(V(U8, 43), I16)
    assert (r.ct, r.get()) == (I16, 43)
    with pytest.raises(eco.NarrowingError):
        eco.value_convert(V(F32, 3.11111), I32)
    r = eco.value_convert(V(F32, 3.11111), F32)
    assert r.ct == F32 and r.get() == np.float32(3.11111)
    r = eco.value_convert(V(U16, 33), F32)
    assert (r.ct, r.get()) == (F32, 33.0)


# ---- B.7 src/value.rs:294-310 ------------------------------------------------
@pytest.mark.parametrize("ct", range(NTYPES))
def test_b7_default_get_f64(ct):
    v = V(ct, 0)
    assert eco.value_convert(v, ct).get() == 0
    r = eco.value_convert(v, F64)
    assert r.ct == F64 and r.get() == 0.0


# ---- B.8 src/ctype.rs:188-207 -------------------------------------------------
def test_b8_union():
    for ct in (U8, U16, F32, F64):
        assert eco.union(ct, ct) == ct
    assert eco.union(I16, F32) == F32 and eco.union(F32, I16) == F32
    assert eco.union(U8, U16) == U16
    assert eco.union(I32, F32) == F64


# SURVEY Appendix A.1/A.2 tables derived from ctype.rs:99-131 (full lattice)
_UNION = """
 u8 u16 u32 u64 i16 i16 i32 i64 f32 f64
u16 u16 u32 u64 i32 i32 i32 i64 f32 f64
u32 u32 u32 u64 i64 i64 i64 i64 f64 f64
u64 u64 u64 u64 f64 f64 f64 f64 f64 f64
i16 i32 i64 f64  i8 i16 i32 i64 f32 f64
i16 i32 i64 f64 i16 i16 i32 i64 f32 f64
i32 i32 i64 f64 i32 i32 i32 i64 f64 f64
i64 i64 i64 f64 i64 i64 i64 i64 f64 f64
f32 f32 f64 f64 f32 f32 f64 f64 f32 f64
f64 f64 f64 f64 f64 f64 f64 f64 f64 f64
"""
_NAMES = ["u8", "u16", "u32", "u64", "i8", "i16", "i32", "i64", "f32", "f64"]


def test_union_table_and_can_fit():
    rows = [r.split() for r in _UNION.strip().splitlines()]
    n_fit = 0
    for a in range(NTYPES):
        for b in range(NTYPES):
            assert eco.union(a, b) == _NAMES.index(rows[a][b]), (a, b)
            assert eco.union(a, b) == eco.union(b, a)
            # unify never fails: a fits into a∪b (value.rs:105-106 unwraps)
            assert eco.can_fit_into(a, eco.union(a, b))
            n_fit += eco.can_fit_into(a, b)
    # 10 identities + 31 widening pairs (SURVEY App. A.2's table has 41 Y; its prose says 40)
    assert n_fit == 41


# ---- B.9 src/ctype.rs:218-243 ---------------------------------------------------
def test_b9_sizes_and_limits():
    sizes = [1, 2, 4, 8, 1, 2, 4, 8, 4, 8]
    for ct in range(NTYPES):
        assert eco.lib().eco_size_of(ct) == sizes[ct]
        dt = np.dtype(NP_DTYPES[ct])
        lo, hi = (np.finfo(dt).min, np.finfo(dt).max) if dt.kind == "f" else (np.iinfo(dt).min, np.iinfo(dt).max)
        assert eco.min_value(ct).get() == lo and eco.max_value(ct).get() == hi
        # zero_one: one + zero == one (ctype.rs:266-278)
        s = eco.value_binop(ADD, V(ct, 1), V(ct, 0))
        assert eco.value_eq(s, V(ct, 1))


# ---- B.11 / B.13 src/buffer.rs:501-513, :567-578 --------------------------------
@pytest.mark.parametrize("ct", range(NTYPES))
def test_b11_b13_convert_buffer(ct):
    buf = np.zeros(3, NP_DTYPES[ct])
    assert eco.convert(buf, ct).dtype == buf.dtype  # to_vec round trip / identity clone
    for target in range(NTYPES):
        if eco.can_fit_into(ct, target):
            r = eco.convert(buf, target)
            assert r.dtype == np.dtype(NP_DTYPES[target]) and r.size == 3
            assert np.array_equal(eco.f_convert(buf, target), r)
        else:
            with pytest.raises(eco.NarrowingError):
                eco.convert(buf, target)


# ---- B.12 src/buffer.rs:516-526 --------------------------------------------------
def test_b12_min_max():
    mn, mx = eco.min_max(np.array([-1.0, 3.0, 2000.0, -5555.5]))
    assert (mn.ct, mn.get(), mx.ct, mx.get()) == (F64, -5555.5, F64, 2000.0)
    mn, mx = eco.min_max(np.array([1, 3, 200, 0], np.uint8))
    assert (mn.ct, mn.get(), mx.ct, mx.get()) == (U8, 0, U8, 200)


# ---- B.14 src/buffer.rs:581-592 ---------------------------------------------------
@pytest.mark.parametrize("ct", range(NTYPES))
def test_b14_buffer_neg(ct):
    buf = np.ones(3, NP_DTYPES[ct])
    r = eco.neg(buf)
    exp = eco.value_neg(one(ct))
    assert r.dtype == np.dtype(NP_DTYPES[exp.ct]) and r[0] == exp.get()
    assert np.array_equal(eco.f_neg(buf), r)


# ---- B.15 src/buffer.rs:595-614 -----------------------------------------------------
@pytest.mark.parametrize("lct", range(NTYPES))
def test_b15_binary_all_pairs(lct):
    lhs_val = one(lct)
    for rct in range(NTYPES):
        rhs_val = eco.value_binop(ADD, one(rct), one(rct))  # rhs_ct.one() + rhs_ct.one() -> Float64(2.0)
        assert rhs_val.ct == F64 and rhs_val.get() == 2.0
        lhs = np.ones(3, NP_DTYPES[lct])
        rhs = np.full(3, 2.0)  # CellBuffer::fill(3, rhs_val): rhs_val is Float64
        for op in OPS:
            for a, av, b, bv in ((lhs, lhs_val, rhs, rhs_val), (rhs, rhs_val, lhs, lhs_val)):
                got = eco.binop(op, a, b)
                exp = eco.value_binop(op, av, bv)
                assert got.dtype == np.float64
                assert all(g == exp.get() for g in got)
        # and with the rhs buffer really typed rct (the loop's evident intent)
        rhs_t = np.full(3, 2, NP_DTYPES[rct])
        for op in OPS:
            got = eco.binop(op, lhs, rhs_t)
            assert got.tolist() == [PYOP[op](1.0, 2.0)] * 3
            got = eco.binop(op, rhs_t, lhs)
            assert got.tolist() == [PYOP[op](2.0, 1.0)] * 3


# ---- B.16 src/buffer.rs:617-621 --------------------------------------------------------
def test_b16_scalar_mul():
    buf = (np.arange(9) + 1).astype(np.uint8)
    r = eco.binop_scalar(MUL, buf, V(F64, 2.0))
    assert r.dtype == np.float64 and r.tolist() == [(i + 1.0) * 2.0 for i in range(9)]


# ---- B.17 src/buffer.rs:624-672 ----------------------------------------------------------
def test_b17_equal_and_cmp():
    buf = np.array([math.nan if i % 2 == 0 else float(i) for i in range(9)])
    assert eco.buffer_cmp(buf, buf) == 0
    z = lambda n, dt: np.zeros(n, dt)
    assert eco.buffer_cmp(z(4, np.uint8), z(4, np.uint8)) == 0
    assert eco.buffer_cmp(z(4, np.uint8), z(5, np.uint8)) != 0
    i32 = lambda *a: np.array(a, np.int32)
    assert eco.buffer_cmp(i32(1, 2, 3), i32(2, 3, 4)) < 0
    assert eco.buffer_cmp(i32(1, 2, 3), i32(2, 3)) < 0
    assert eco.buffer_cmp(np.array([math.nan, 2.0, 3.0]), np.array([math.nan, 2.0, 4.0])) < 0
    assert eco.buffer_cmp(z(4, np.uint8), z(4, np.float32)) < 0
    assert eco.buffer_cmp(z(4, np.float32), z(4, np.uint8)) > 0
    assert eco.buffer_cmp(z(4, np.uint8), z(5, np.uint8)) < 0
    assert eco.buffer_cmp(z(5, np.uint8), z(4, np.uint8)) > 0
    assert eco.buffer_cmp(z(4, np.float64), z(5, np.float64)) < 0
    assert eco.buffer_cmp(z(5, np.float64), z(4, np.float64)) > 0


# ---- B.18 src/masked/mask.rs:184-242 --------------------------------------------------------
def test_b18_mask():
    t3, f3 = np.ones(3, np.uint8), np.zeros(3, np.uint8)
    assert eco.mask_counts(t3) == (3, 0) and eco.mask_counts(f3) == (0, 3)
    assert eco.mask_counts(np.array([1, 0, 1], np.uint8)) == (2, 1)
    t4, f4 = np.ones(4, np.uint8), np.zeros(4, np.uint8)
    assert np.array_equal(eco.mask_not(t4), f4)
    m = np.array([1, 0, 1, 0], np.uint8)
    assert np.array_equal(eco.mask_not(m), 1 - m)
    assert not eco.mask_all(m, True) and not eco.mask_all(m, False)
    assert eco.mask_all(t4, True) and not eco.mask_all(t4, False)
    l, r = m, 1 - m
    assert eco.mask_all(eco.mask_and(l, r), False)
    assert eco.mask_all(eco.mask_or(l, r), True)
    # borrowed forms zip to the shorter operand (mask.rs:129-140)
    assert eco.mask_and(np.ones(5, np.uint8), np.ones(3, np.uint8)).size == 3


# ---- B.19 src/masked/nodata.rs:75-95 -----------------------------------------------------------
def test_b19_nodata():
    assert eco.nodata_value(ND_NONE, I16) is None
    assert eco.nodata_value(ND_DEFAULT, U8).get() == 0
    assert math.isnan(eco.nodata_value(ND_DEFAULT, F32).get())
    assert eco.nodata_value(ND_VALUE, U16, V(U16, 6)).get() == 6
    for ct in range(NTYPES):
        assert eco.nodata_value(ND_DEFAULT, ct) is not None
    # f64::NAN.is(NoData::<f64>::Default)
    assert eco.mask_from_nodata(np.array([math.nan]), ND_DEFAULT).tolist() == [0]
    # Rust's NAN constants
    assert eco.nodata_value(ND_DEFAULT, F64).bits() == 0x7FF8000000000000
    assert eco.nodata_value(ND_DEFAULT, F32).bits() == 0x7FC00000
    assert eco.nodata_value(ND_DEFAULT, I16).get() == -32768


# ---- B.20 src/masked/masked_buffer.rs:413-425 ------------------------------------------------------
def test_b20_vec_with_nodata():
    v = np.array([1.0, math.nan, 3.0, math.nan])
    assert eco.mask_from_nodata(v, ND_DEFAULT).tolist() == [1, 0, 1, 0]
    assert eco.mask_from_nodata(v, ND_VALUE, V(F64, 3.0)).tolist() == [1, 1, 0, 1]
    assert eco.mask_from_nodata(v, ND_NONE).tolist() == [1, 1, 1, 1]
    assert np.array_equal(eco.f_mask_from_nodata(v, eco.nodata_value(ND_DEFAULT, F64)), [1, 0, 1, 0])


# ---- B.21 src/masked/masked_buffer.rs:465-479 --------------------------------------------------------
def test_b21_masked_neg_to_vec_with_nodata():
    buf = np.arange(9, dtype=np.uint8)
    mask = (np.arange(9) % 2 == 0).astype(np.uint8)
    r = eco.neg(buf)
    assert r.dtype == np.int16
    r = eco.convert(r, I16)
    v = eco.mask_select(r, mask, ND_DEFAULT)
    m = -32768
    assert v.tolist() == [0, m, -2, m, -4, m, -6, m, -8]


# ---- B.22 src/masked/masked_buffer.rs:482-509 -----------------------------------------------------------
def test_b22_masked_min_max_and_scalar():
    buf = np.arange(9, dtype=np.uint8)
    mask = np.array([i != 0 and i != 8 for i in range(9)], np.uint8)
    mn, mx = eco.min_max(buf, mask)
    assert (mn.ct, mn.get(), mx.ct, mx.get()) == (U8, 1, U8, 7)
    r = eco.binop_scalar(MUL, buf, V(F64, 2.0))
    mask = (np.arange(9) % 2 == 0).astype(np.uint8)
    fmin = np.finfo(np.float64).min
    v = eco.mask_select(eco.convert(r, F64), mask, ND_VALUE, V(F64, fmin))
    assert v.tolist() == [0.0, fmin, 4.0, fmin, 8.0, fmin, 12.0, fmin, 16.0]
    # fully masked -> inverted sentinels (App. A.5)
    mn, mx = eco.min_max(buf, np.zeros(9, np.uint8))
    assert (mn.get(), mx.get()) == (255, 0)


# ---- B.23 src/masked/masked_buffer.rs:512-531, :443-447, examples/masked.rs:5-23 ---------------------------
def test_b23_masked_binary_and_example():
    lhs, lmask = np.full(9, 1.0), (np.arange(9) % 2 == 0).astype(np.uint8)
    rhs, rmask = np.full(9, 2.0), np.ones(9, np.uint8)
    for op in OPS:
        r = eco.binop(op, lhs, rhs)
        m = eco.mask_and(lmask, rmask)
        assert r[0] == PYOP[op](1.0, 2.0) and m[0] == 1 and m[1] == 0
    buf = np.arange(4, dtype=np.float64)
    mask = np.array([1, 0, 1, 0], np.uint8)
    assert eco.mask_counts(mask) == (2, 2)
    r = eco.binop_scalar(MUL, eco.binop(ADD, buf, np.ones(4)), V(F64, 2.0))
    assert r.tolist() == [2.0, 4.0, 6.0, 8.0]
    assert eco.mask_and(mask, np.ones(4, np.uint8)).tolist() == [1, 0, 1, 0]
    assert eco.convert(np.arange(4, dtype=np.uint8), F64).tolist() == [0.0, 1.0, 2.0, 3.0]


# ---- empty-result quirk src/buffer.rs:233-234 -----------------------------------------------------------------
def test_empty_results_are_uint8():
    e = np.zeros(0, np.uint16)
    assert eco.binop(ADD, e, e).dtype == np.uint8
    assert eco.neg(e).dtype == np.uint8
    assert eco.convert(e, F32).dtype == np.uint8
    assert eco.convert(e, U16).dtype == np.uint16  # identity is a clone (buffer.rs:151-153)
    # zip truncation (buffer.rs:327)
    assert eco.binop(ADD, np.ones(5, np.uint8), np.ones(3, np.int64)).size == 3


# ---- B.24-B.27 src/gdal/rasterband.rs:20-36,57-71,138-191 -----------------------------------------------------
NDVI_MIN_HEX, NDVI_MAX_HEX = "-0x1.ff8ca5bcc77dcp-4", "0x1.5708125b0ed28p-1"


def _bands(golden_dir):
    red, _ = read_tiff(os.path.join(golden_dir, "L8-Elkton-VA-B4.tiff"))
    nir, _ = read_tiff(os.path.join(golden_dir, "L8-Elkton-VA-B5.tiff"))
    nir_nd, nd = read_tiff(os.path.join(golden_dir, "L8-Elkton-VA-B5-nd.tiff"))
    return red.ravel(), nir.ravel(), nir_nd.ravel(), nd


def test_b24_band_min_max(golden_dir):
    red, nir, nir_nd, nd = _bands(golden_dir)
    assert nd == 0.0
    for band, (lo, hi) in ((nir, (5469, 39368)), (red, (6396, 27835)), (nir_nd, (0, 39368))):
        mn, mx = eco.min_max(band)
        assert (mn.ct, mn.get(), mx.ct, mx.get()) == (U16, lo, U16, hi)
        assert (mn.get(), mx.get()) == (band.min(), band.max())


def test_b25_ndvi_unmasked(golden_dir):
    red, nir, _, _ = _bands(golden_dir)
    ndvi = eco.binop(DIV, eco.binop(SUB, nir, red), eco.binop(ADD, nir, red))
    mn, mx = eco.min_max(ndvi)
    # the reference's one-sided tolerance (rasterband.rs:161-162) and its quoted gdal_calc values
    assert mn.get() - -0.1248899911993 < 1e-8 and abs(mn.get() - -0.1248899911993) < 1e-8
    assert mx.get() - 0.66998345719859 < 1e-8 and abs(mx.get() - 0.66998345719859) < 1e-8
    assert float(mn.get()).hex() == NDVI_MIN_HEX and float(mx.get()).hex() == NDVI_MAX_HEX
    assert abs(ndvi.mean() - 0.45559234941397) < 1e-12  # quoted, not asserted, by the reference


def test_b26_b27_ndvi_masked(golden_dir):
    red, _, nir, nd = _bands(golden_dir)
    ndv = Value.of(U16, int(nd))
    rmask = eco.mask_from_nodata(red, ND_VALUE, ndv)
    nmask = eco.mask_from_nodata(nir, ND_VALUE, ndv)
    assert eco.mask_counts(nmask) == (31430, 4)
    assert sum(eco.mask_counts(nmask)) == 186 * 169
    assert np.flatnonzero(nmask == 0).tolist() == [14116, 17508, 21136, 29930]
    num, m1 = eco.binop(SUB, nir, red), eco.mask_and(nmask, rmask)
    den, m2 = eco.binop(ADD, nir, red), eco.mask_and(nmask, rmask)
    ndvi, m = eco.binop(DIV, num, den), eco.mask_and(m1, m2)
    assert eco.mask_counts(m) == eco.mask_counts(nmask)
    mn, mx = eco.min_max(ndvi, m)
    assert float(mn.get()).hex() == NDVI_MIN_HEX and float(mx.get()).hex() == NDVI_MAX_HEX
    # masked-out cells are still computed (masked_buffer.rs:331): (0 - red)/(0 + red) == -1
    assert all(ndvi[i] == -1.0 for i in np.flatnonzero(m == 0))
    # config 5: convert(Float32) first, then the same arithmetic (f32 operands widen exactly)
    ndvi32 = eco.binop(DIV, eco.binop(SUB, eco.convert(nir, F32), eco.convert(red, F32)),
                       eco.binop(ADD, eco.convert(nir, F32), eco.convert(red, F32)))
    assert np.array_equal(ndvi32, ndvi)
