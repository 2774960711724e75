"""The reference's own unit tests, doc-tests and examples,
restated against the HIP path through the host mirror.  Each test names the
reference test it mirrors; the same known answers pin the oracle in
tests/test_oracle_kat.py, so HIP == oracle == reference KATs.
"""
import math
import os

import numpy as np
import pytest

from tiff_util import read_tiff
from vectors import bits_of, rand_cells

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ec():
    import erased_cells_hip as ec
    ec.init(0)
    return ec


def test_quick_example(ec):
    """examples/quick.rs:5-11"""
    buf1 = ec.CellBuffer.from_vec(np.array([1, 2, 3], np.uint8))
    buf2 = ec.CellBuffer.from_vec(np.array([2, 4, 6], np.uint16))
    result = buf1 / buf2 * 0.5
    assert result == ec.CellBuffer.from_vec([0.25, 0.25, 0.25])
    assert result.cell_type() == ec.Float64


def test_buffer_example(ec):
    """examples/buffer.rs:5-30 and the doc-test at src/buffer.rs:20-49"""
    buf1 = ec.CellBuffer.fill_via(9, lambda i: i, np.uint8)
    assert buf1.cell_type() == ec.UInt8
    assert buf1.get(3) == ec.CellValue(ec.UInt8, 3)
    mn, mx = buf1.min_max()
    assert (mn, mx) == (ec.CellValue(ec.UInt8, 0), ec.CellValue(ec.UInt8, 8))
    assert (mn.ct, mx.ct) == (ec.UInt8, ec.UInt8)
    buf2 = ec.CellBuffer.fill_via(9, lambda i: 8.0 - i, np.float32)
    assert buf2.cell_type() == ec.Float32
    assert buf2.min_max() == (ec.CellValue(ec.Float32, 0.0), ec.CellValue(ec.Float32, 8.0))
    diff = buf2 - buf1
    assert diff.min_max() == (ec.CellValue.new(-8), ec.CellValue.new(8))


def test_buffer_defaults_put_get(ec):
    """src/buffer.rs:462-487 (defaults, put_get)"""
    for ct in ec.CELL_TYPES:
        cv = ec.CellBuffer.with_defaults(3, ct)
        assert cv.len() == 3 and cv.cell_type() == ct
        assert cv.get(0) == ec.CellValue(ct, 0)
        cv = ec.CellBuffer.fill(3, ec.CellValue(ct, 0))
        one = ec.CellValue(ct, 1)
        cv.put(1, one)
        assert cv.get(1) == one.convert(ct)
    with pytest.raises(ec.NarrowingError):
        ec.CellBuffer.with_defaults(3, ec.UInt8).put(0, ec.CellValue(ec.Float32, 1.5))
    with pytest.raises(IndexError):
        ec.CellBuffer.with_defaults(3, ec.UInt8).get(3)


def test_buffer_and_masked_extend(ec):
    """src/buffer.rs:489-498 (extend) and src/masked/masked_buffer.rs:449-455"""
    buf = ec.CellBuffer.fill(3, ec.CellValue(ec.UInt8, 0))
    assert not buf.is_empty() and buf.cell_type() == ec.UInt8
    buf.extend([1])
    assert buf.cell_type() == ec.UInt8 and buf.len() == 4
    assert buf.get(0) == ec.CellValue.new(0) and buf.get(3) == ec.CellValue.new(1)
    with pytest.raises(OverflowError):
        buf.extend([300])  # to_u8() is None -> unwrap panics in the reference
    m = ec.MaskedCellBuffer.fill(3, ec.CellValue.new(0))
    m.extend([(1, False)])
    assert m.get_masked(0) == ec.CellValue.new(0) and m.get_masked(3) is None and m.len() == 4
    e = ec.CellBuffer.empty(0, ec.Float32)
    e.extend([1.5, 2])
    assert e.to_numpy().tolist() == [1.5, 2.0] and e.cell_type() == ec.Float32


def test_lazy_operator_syntax_runs_fused_and_matches_eager(ec):
    """`lazy()` defers the reference's operator chains (src/gdal/rasterband.rs:148, examples/masked.rs:12) so that they
    run as fused launches; every tree shape gives the eager chain's bits."""
    lz = ec.fused.lazy
    a = ec.CellBuffer.from_vec(rand_cells(ec.UInt16, 5000, 11))
    b = ec.CellBuffer.from_vec(rand_cells(ec.UInt16, 5000, 12))
    c = ec.CellBuffer.from_vec(rand_cells(ec.Float32, 5000, 13))
    same = lambda x, y: x.cell_type() == y.cell_type() and np.array_equal(bits_of(x.to_numpy()), bits_of(y.to_numpy()))
    A, Bz, Cz = lz(a), lz(b), lz(c)
    assert same(((A - Bz) / (A + Bz)).eval(), (a - b) / (a + b))               # (x o1 y) o2 (z o3 w)
    assert same(((A + Bz) * 2.0).eval(), (a + b) * 2.0)                          # (x o1 y) o2 scalar
    assert same(((A + Bz) * Cz).eval(), (a + b) * c)                             # mixed cell types
    assert same((A / Bz).eval(), a / b)                                          # a single op stays a single op
    assert same((Cz * ((A - Bz) / (A + Bz)) + 1).eval(), c * ((a - b) / (a + b)) + 1)    # deeper: sub-trees first
    assert same((Cz / (A + Bz)).eval(), c / (a + b))                             # node on the right only
    assert same((2.0 * A).eval(), a * 2.0)                                       # scalar on the left
    one_minus = (1 - A).eval().to_numpy()
    assert np.array_equal(bits_of(one_minus), bits_of(1.0 - a.to_numpy().astype(np.float64)))
    ma = ec.MaskedCellBuffer.from_vec_with_nodata(rand_cells(ec.UInt16, 5000, 14), ec.NoData.new(0))
    mb = ec.MaskedCellBuffer.from_vec(rand_cells(ec.UInt16, 5000, 15))
    got, exp = ((lz(ma) + lz(mb)) * 2.0).eval(), (ma + mb) * 2.0
    assert same(got.buffer(), exp.buffer()) and got.mask() == exp.mask()
    assert (lz(3) + 4).eval() == ec.CellValue(ec.Float64, 7.0)


def test_from_others_and_iterators(ec):
    """src/buffer.rs:528-555 (from_others), src/masked/masked_buffer.rs:457-462 (from_iter), the IntoIterator impls."""
    b = ec.CellBuffer.from_values([ec.CellValue(ec.UInt16, x) for x in (3, 4, 5)])
    assert b.cell_type() == ec.UInt16 and b.len() == 3 and b.get(2) == ec.CellValue(ec.UInt16, 5)
    v = np.array([33.3, 44.4, 55.5], np.float32)
    b = ec.CellBuffer.from_values([ec.CellValue.new(x) for x in v])
    assert b.cell_type() == ec.Float32 and b.len() == 3 and b.get(2) == ec.CellValue(ec.Float32, np.float32(55.5))
    b = ec.CellBuffer.from_vec(v)
    assert b.cell_type() == ec.Float32 and b.len() == 3 and b.get(2) == ec.CellValue(ec.Float32, np.float32(55.5))
    assert ec.CellBuffer.from_values([]).cell_type() == ec.UInt8
    # the first value decides the type; later values go through get::<T>().unwrap()
    mixed = ec.CellBuffer.from_values([ec.CellValue(ec.Int32, 7), ec.CellValue(ec.UInt8, 9)])
    assert mixed.cell_type() == ec.Int32 and mixed.to_numpy().tolist() == [7, 9]
    with pytest.raises(ec.NarrowingError):
        ec.CellBuffer.from_values([ec.CellValue(ec.UInt8, 7), ec.CellValue(ec.Int32, 9)])
    assert list(b) == [ec.CellValue(ec.Float32, x) for x in v]
    buf = ec.MaskedCellBuffer.from_iter(range(5), dtype=np.int16)
    assert buf.mask().all(True) and buf.to_vec(ec.Int16).tolist() == [0, 1, 2, 3, 4]
    pairs = ec.MaskedCellBuffer.from_iter([(1, True), (2, False)], dtype=np.uint8)
    assert list(pairs) == [(ec.CellValue(ec.UInt8, 1), True), (ec.CellValue(ec.UInt8, 2), False)]
    m = ec.Mask.new([True, False, True])
    assert list(m) == [True, False, True] and m[1] is False
    m[1] = True
    assert m.all(True)


def test_debug_rendering(ec):
    """src/lib.rs:197-206 (elided), src/buffer.rs:558-564 and src/masked/masked_buffer.rs:533-540 (debug)."""
    from erased_cells_hip.buffer import elided, rust_debug
    assert elided([1] * 3) == "1, 1, 1"
    assert elided([0] * 30) == "0, 0, 0, 0, 0, ... 0, 0, 0, 0, 0"
    b = ec.CellBuffer.fill(5, 37)
    assert repr(b).startswith("Int32CellBuffer") and repr(b) == "Int32CellBuffer(37, 37, 37, 37, 37)"
    b = ec.CellBuffer.fill(15, 37)
    assert "..." in repr(b) and repr(b) == "Int32CellBuffer(37, 37, 37, 37, 37, ... 37, 37, 37, 37, 37)"
    m = ec.MaskedCellBuffer.from_vec(np.arange(0, 1, dtype=np.int32))
    dbg = repr(m)
    assert dbg.startswith("Int32MaskedCellBuffer") and "CellBuffer(0)" in dbg and "Mask(true)" in dbg
    assert dbg == "Int32MaskedCellBuffer(Int32CellBuffer(0), Mask(true))"
    # Rust `{:?}` of floats: shortest round-trip digits, `.0` on integral values, scientific outside [1e-4, 1e16)
    got = [rust_debug(np.float64(x)) for x in (0.25, 37.0, -0.0, 1e-7, 1e16, 1.5e300, 123456789012345680.0, 0.0001, 0.00001)]
    assert got == ["0.25", "37.0", "-0.0", "1e-7", "1e16", "1.5e300", "1.2345678901234568e17", "0.0001", "1e-5"]
    assert [rust_debug(np.float32(x)) for x in (0.1, 16777216.0, 1e-10)] == ["0.1", "16777216.0", "1e-10"]
    assert [rust_debug(x) for x in (np.float64("nan"), np.float32("inf"), -np.float64("inf"))] == ["NaN", "inf", "-inf"]
    assert repr(ec.CellBuffer.from_vec(np.array([0.5, 2.0], np.float32))) == "Float32CellBuffer(0.5, 2.0)"
    assert repr(ec.CellValue.new(37)) == "Int32(37)" and repr(ec.CellBuffer.empty(0, ec.UInt8)) == "UInt8CellBuffer()"


def test_masked_buffer_ordering(ec):
    """derived PartialOrd on MaskedCellBuffer(CellBuffer, Mask) (masked_buffer.rs:39): lexicographic over the pair."""
    a = ec.MaskedCellBuffer(ec.CellBuffer.from_vec(np.array([1, 2, 3], np.uint8)), ec.Mask.new([True, False, True]))
    b = ec.MaskedCellBuffer(ec.CellBuffer.from_vec(np.array([1, 2, 3], np.uint8)), ec.Mask.new([True, True, True]))
    c = ec.MaskedCellBuffer(ec.CellBuffer.from_vec(np.array([1, 2, 4], np.uint8)), ec.Mask.new([False, False, False]))
    assert a.cmp(a) == 0 and a < b and b > a and b < c and a < c and not (c < a)
    assert a != b and a == ec.MaskedCellBuffer(a.buffer().clone(), a.mask().clone())


def test_serde_wire_shapes_round_trip(ec):
    """serde's externally tagged shapes (parity unpinned — the reference holds no serialized
    fixture; see erased_cells_hip/wire.py)."""
    w = ec.wire
    assert w.dumps(ec.CellBuffer.from_vec(np.array([1, 2, 3], np.uint8))) == '{"UInt8":[1,2,3]}'
    assert w.dumps(ec.CellBuffer.from_vec(np.array([1.5, -2.0], np.float32))) == '{"Float32":[1.5,-2.0]}'
    # an f32 prints with the shortest digits that round-trip as f32, as serde_json does (not 0.10000000149011612)
    assert w.dumps(ec.CellBuffer.from_vec(np.array([0.1, 1e20, 3.0e-7], np.float32))) == '{"Float32":[0.1,1e20,3e-7]}'
    assert w.dumps(ec.CellValue.new(np.float32(0.1))) == '{"Float32":0.1}'
    assert w.dumps(ec.Mask.new([True, False])) == "[true,false]"
    m = ec.MaskedCellBuffer.from_vec_with_nodata(np.array([0, 7, 0, 9], np.uint16), ec.NoData.new(0))
    assert w.dumps(m) == '[{"UInt16":[0,7,0,9]},[false,true,false,true]]'
    assert w.dumps(ec.CellValue.new(np.int64(-5))) == '{"Int64":-5}'
    assert [w.to_wire(x) for x in (ec.NoData.none(), ec.NoData.default(), ec.NoData.new(3))] == ["None", "Default", {"Value": 3}]
    assert w.cell_type_to_wire(ec.Float64) == "Float64" and w.cell_type_from_wire("Int8") == ec.Int8
    for ct in ec.CELL_TYPES:
        a = rand_cells(ct, 257, 900 + ct)
        if ec.NP_DTYPES[ct].kind == "f":
            a = np.where(np.isfinite(a), a, 1.0).astype(a.dtype)  # non-finite cells serialize as null
        b = w.loads_buffer(w.dumps(ec.CellBuffer.from_vec(a)))
        assert b.cell_type() == ct and np.array_equal(bits_of(b.to_numpy()), bits_of(a))
    back = w.loads_masked(w.dumps(m))
    assert back.buffer() == m.buffer() and back.mask() == m.mask()
    assert w.to_wire(ec.CellBuffer.from_vec(np.array([np.nan, np.inf, 1.0])))["Float64"] == [None, None, 1.0]
    for bad in ('{"UInt8":[256]}', '{"UInt8":[1.5]}', '{"Float64":[null]}', '{"UInt57":[1]}', '{"UInt8":[1],"Int8":[1]}'):
        with pytest.raises(ValueError):
            w.loads_buffer(bad)
    with pytest.raises(ValueError):
        w.loads_masked('[{"UInt8":[1]}]')
    with pytest.raises(AssertionError):
        w.loads_masked('[{"UInt8":[1,2]},[true]]')  # MaskedCellBuffer::new length assert (masked_buffer.rs:48-53)
    assert w.loads_buffer('{"Int32":[]}').cell_type() == ec.Int32


def test_integer_literals_default_to_i32_and_refuse_overflow(ec):
    """Python ints stand for Rust integer literals, which are i32 unless annotated: out-of-range ones do not compile
    in Rust and raise here instead of wrapping."""
    assert ec.CellValue.new(7).cell_type() == ec.Int32
    assert ec.CellBuffer.from_vec([1, 2, 3]).cell_type() == ec.Int32
    for bad in (2**31, -2**31 - 1):
        with pytest.raises(OverflowError):
            ec.CellValue.new(bad)
        with pytest.raises(OverflowError):
            ec.CellBuffer.from_vec([1, bad])
    assert ec.CellValue.new(np.int64(2**40)).cell_type() == ec.Int64


def test_buffer_to_vec(ec):
    """src/buffer.rs:501-513"""
    for ct in ec.CELL_TYPES:
        v = np.zeros(3, ec.NP_DTYPES[ct])
        assert np.array_equal(ec.CellBuffer.from_vec(v).to_vec(ct), v)


def test_buffer_min_max(ec):
    """src/buffer.rs:516-526"""
    mn, mx = ec.CellBuffer.from_vec([-1.0, 3.0, 2000.0, -5555.5]).min_max()
    assert (mn, mx) == (ec.CellValue(ec.Float64, -5555.5), ec.CellValue(ec.Float64, 2000.0))
    mn, mx = ec.CellBuffer.from_vec(np.array([1, 3, 200, 0], np.uint8)).min_max()
    assert (mn.ct, mn.value, mx.ct, mx.value) == (ec.UInt8, 0, ec.UInt8, 200)


def test_buffer_convert(ec):
    """src/buffer.rs:567-578"""
    for ct in ec.CELL_TYPES:
        buf = ec.CellBuffer.with_defaults(3, ct)
        for target in (t for t in ec.CELL_TYPES if ec.can_fit_into(ct, t)):
            r = buf.convert(target)
            assert r.cell_type() == target


def test_buffer_unary(ec):
    """src/buffer.rs:581-592 with the result types of src/value.rs:338-346"""
    exp = {ec.UInt8: ec.Int16, ec.UInt16: ec.Int32, ec.UInt32: ec.Float64, ec.UInt64: ec.Float64}
    for ct in ec.CELL_TYPES:
        buf = -ec.CellBuffer.fill(3, ec.CellValue(ct, 1))
        assert buf.cell_type() == exp.get(ct, ct)
        assert buf.get(0) == ec.CellValue.new(-1)


def test_buffer_binary(ec):
    """src/buffer.rs:595-614: all 100 type pairs, four ops, both orders"""
    pyop = [lambda a, b: a + b, lambda a, b: a - b, lambda a, b: a * b, lambda a, b: a / b]
    for lhs_ct in ec.CELL_TYPES:
        for rhs_ct in ec.CELL_TYPES:
            lhs = ec.CellBuffer.fill(3, ec.CellValue(lhs_ct, 1))
            rhs = ec.CellBuffer.fill(3, ec.CellValue(rhs_ct, 2))
            for op in range(4):
                r = lhs._binop(op, rhs)
                assert r.cell_type() == ec.Float64
                assert r.to_numpy().tolist() == [pyop[op](1.0, 2.0)] * 3
                assert rhs._binop(op, lhs).to_numpy().tolist() == [pyop[op](2.0, 1.0)] * 3


def test_buffer_scalar(ec):
    """src/buffer.rs:617-621"""
    buf = ec.CellBuffer.fill_via(9, lambda i: i + 1, np.uint8)
    r = buf * 2.0
    assert r == ec.CellBuffer.fill_via(9, lambda i: (i + 1.0) * 2.0, np.float64)


def test_buffer_equal_cmp(ec):
    """src/buffer.rs:624-672"""
    buf = ec.CellBuffer.fill_via(9, lambda i: math.nan if i % 2 == 0 else float(i), np.float64)
    assert buf == buf
    wd = ec.CellBuffer.with_defaults
    assert wd(4, ec.UInt8) == wd(4, ec.UInt8)
    assert wd(4, ec.UInt8) != wd(5, ec.UInt8)
    assert ec.CellBuffer.from_vec([1, 2, 3]) < ec.CellBuffer.from_vec([2, 3, 4])
    assert ec.CellBuffer.from_vec([1, 2, 3]) < ec.CellBuffer.from_vec([2, 3])
    assert ec.CellBuffer.from_vec([math.nan, 2.0, 3.0]) < ec.CellBuffer.from_vec([math.nan, 2.0, 4.0])
    assert wd(4, ec.UInt8) < wd(4, ec.Float32) and wd(4, ec.Float32) > wd(4, ec.UInt8)
    assert wd(4, ec.UInt8) < wd(5, ec.UInt8) and wd(5, ec.UInt8) > wd(4, ec.UInt8)
    assert wd(4, ec.Float64) < wd(5, ec.Float64) and wd(5, ec.Float64) > wd(4, ec.Float64)


def test_mask_tests(ec):
    """src/masked/mask.rs:184-242 (counts, set, not, all, and, or)"""
    assert ec.Mask.fill(3, True).counts() == (3, 0)
    assert ec.Mask.fill(3, False).counts() == (0, 3)
    assert ec.Mask.fill_via(3, lambda i: i % 2 == 0).counts() == (2, 1)
    m = ec.Mask.fill(3, True)
    m.put(1, False)
    m.put(0, False)
    assert m == ec.Mask.new([False, False, True])
    t, f = ec.Mask.fill(4, True), ec.Mask.fill(4, False)
    assert ~t == f
    m, r = ec.Mask.new([True, False, True, False]), ec.Mask.new([False, True, False, True])
    assert ~m == r
    m = ec.Mask.fill_via(4, lambda i: i % 2 == 0)
    assert not m.all(True) and not m.all(False)
    assert ec.Mask.fill(4, True).all(True) and not ec.Mask.fill(4, True).all(False)
    l, r = ec.Mask.fill_via(4, lambda i: i % 2 == 0), ec.Mask.fill_via(4, lambda i: i % 2 != 0)
    assert (l & r).all(False) and (l | r).all(True)


def test_nodata_tests(ec):
    """src/masked/nodata.rs:75-95"""
    assert ec.NoData.none().value(ec.Int16) is None
    assert ec.NoData.default().value(ec.UInt8).value == 0
    assert math.isnan(ec.NoData.default().value(ec.Float32).value)
    assert ec.NoData.new(np.uint16(6)).value(ec.UInt16).value == 6
    for ct in ec.CELL_TYPES:
        assert ec.NoData.default().value(ct) is not None
    m = ec.MaskedCellBuffer.from_vec_with_nodata([math.nan], ec.NoData.default())
    assert m.mask().to_numpy().tolist() == [0]


def test_masked_ctor_and_vec_with_nodata(ec):
    """src/masked/masked_buffer.rs:400-425"""
    m = ec.MaskedCellBuffer.fill_via(3, lambda i: i, np.uint8)
    r = ec.MaskedCellBuffer.new(ec.CellBuffer.fill_via(3, lambda i: i, np.uint8), ec.Mask.fill(3, True))
    assert m == r
    assert ec.MaskedCellBuffer.from_vec([0.0] * 4).mask().counts() == (4, 0)
    assert ec.MaskedCellBuffer.with_defaults(4, ec.Int16).mask().counts() == (4, 0)
    v = [1.0, math.nan, 3.0, math.nan]
    m = ec.MaskedCellBuffer.from_vec_with_nodata(v, ec.NoData.default())
    assert m == ec.MaskedCellBuffer.new(ec.CellBuffer.from_vec(v), ec.Mask.new([True, False, True, False]))
    m = ec.MaskedCellBuffer.from_vec_with_nodata(v, ec.NoData.new(3.0))
    assert m == ec.MaskedCellBuffer.new(ec.CellBuffer.from_vec(v), ec.Mask.new([True, True, False, True]))
    with pytest.raises(AssertionError):
        ec.MaskedCellBuffer.new(ec.CellBuffer.from_vec(v), ec.Mask.fill(3, True))


def _filler_masker(i):
    return (i, i % 2 == 0)


def test_masked_get_masked_convert(ec):
    """src/masked/masked_buffer.rs:428-447"""
    buf = ec.MaskedCellBuffer.fill_with_mask_via(9, _filler_masker, np.uint8)
    assert buf.get(4) == ec.CellValue.new(4)
    assert buf.get_masked(4) == ec.CellValue.new(4)
    assert buf.get_masked(5) is None
    buf.put(5, ec.CellValue(ec.UInt8, 4))
    assert buf.get_masked(5) is None
    buf.mask().put(5, True)
    assert buf.get_masked(5) == ec.CellValue.new(4)
    buf.put_with_mask(5, ec.CellValue(ec.UInt8, 99), False)
    assert buf.get_masked(5) is None
    buf = ec.MaskedCellBuffer.fill_with_mask_via(4, _filler_masker, np.uint8)
    assert buf.convert(ec.Float64).to_vec(ec.Float64).tolist() == [0.0, 1.0, 2.0, 3.0]


def test_masked_unary(ec):
    """src/masked/masked_buffer.rs:465-479"""
    mbuf = ec.MaskedCellBuffer.fill_with_mask_via(9, _filler_masker, np.uint8)
    r = -mbuf
    v = r.to_vec_with_nodata(ec.Int16, ec.NoData.default())
    m = -32768
    assert v.dtype == np.int16 and v.tolist() == [0, m, -2, m, -4, m, -6, m, -8]


def test_masked_min_max_and_scalar(ec):
    """src/masked/masked_buffer.rs:482-509"""
    mbuf = ec.MaskedCellBuffer.fill_with_mask_via(9, lambda i: (i, i != 0 and i != 8), np.uint8)
    assert mbuf.min_max() == (ec.CellValue(ec.UInt8, 1), ec.CellValue(ec.UInt8, 7))
    mbuf = ec.MaskedCellBuffer.fill_with_mask_via(9, lambda i: (i, True), np.uint8)
    r = mbuf * 2.0
    expected = ec.CellBuffer.fill_via(9, lambda i: i, np.uint8) * 2.0
    assert r == ec.MaskedCellBuffer.from_buffer(expected)
    mbuf = ec.MaskedCellBuffer.fill_with_mask_via(9, _filler_masker, np.uint8)
    r = mbuf * 2.0
    assert r != ec.MaskedCellBuffer.from_buffer(expected)
    fmin = np.finfo(np.float64).min
    v = r.to_vec_with_nodata(ec.Float64, ec.NoData.new(fmin))
    assert v.tolist() == [0.0, fmin, 4.0, fmin, 8.0, fmin, 12.0, fmin, 16.0]


def test_masked_binary(ec):
    """src/masked/masked_buffer.rs:512-531"""
    lhs = ec.MaskedCellBuffer.new(ec.CellBuffer.fill(9, 1.0), ec.Mask.fill_via(9, lambda i: i % 2 == 0))
    rhs = ec.MaskedCellBuffer.new(ec.CellBuffer.fill(9, 2.0), ec.Mask.fill(9, True))
    pyop = [lambda a, b: a + b, lambda a, b: a - b, lambda a, b: a * b, lambda a, b: a / b]
    for op in range(4):
        r = lhs._binop(op, rhs)
        assert r.get_masked(0) == ec.CellValue.new(pyop[op](1.0, 2.0))
        assert r.get_masked(1) is None
        assert r.get_masked(4) == ec.CellValue.new(pyop[op](1.0, 2.0))
        assert r.get_masked(5) is None


def test_masked_example(ec):
    """examples/masked.rs:5-23 and the doc-test at src/masked/masked_buffer.rs:15-38"""
    buf = ec.MaskedCellBuffer.fill_with_mask_via(4, lambda i: (float(i), i % 2 == 0), np.float64)
    assert buf.mask() == ec.Mask.new([True, False, True, False])
    assert buf.counts() == (2, 2)
    ones = ec.MaskedCellBuffer.from_vec([1.0] * 4)
    r = (buf + ones) * 2.0
    expected = ec.MaskedCellBuffer.new(ec.CellBuffer.from_vec([2.0, 4.0, 6.0, 8.0]), ec.Mask.new([True, False, True, False]))
    assert r == expected


# ---- GDAL tests on the Landsat fixtures: src/gdal/rasterband.rs:138-191, doc-tests :20-36, :57-71
NDVI_MIN_HEX, NDVI_MAX_HEX = "-0x1.ff8ca5bcc77dcp-4", "0x1.5708125b0ed28p-1"


def _band(golden_dir, name):
    cells, nd = read_tiff(os.path.join(golden_dir, f"L8-Elkton-VA-{name}.tiff"))
    return cells, nd


def test_read_cells_min_max_doc_test(ec, golden_dir):
    """src/gdal/rasterband.rs:27-33: buffer.min_max() equals the band's min/max (B.24)"""
    for name, lo, hi in (("B5", 5469, 39368), ("B4", 6396, 27835), ("B5-nd", 0, 39368)):
        cells, _ = _band(golden_dir, name)
        mn, mx = ec.CellBuffer.from_vec(cells.ravel()).min_max()
        assert (mn.ct, mn.value, mx.value) == (ec.UInt16, lo, hi)


def test_read_cells_ndvi(ec, golden_dir):
    """src/gdal/rasterband.rs:138-163 (B.25)"""
    red = ec.CellBuffer.from_vec(_band(golden_dir, "B4")[0].ravel())
    nir = ec.CellBuffer.from_vec(_band(golden_dir, "B5")[0].ravel())
    ndvi = (nir - red) / (nir + red)
    mn, mx = ndvi.min_max()
    assert mn.to_f64() - -0.1248899911993 < 1e-8 and mx.to_f64() - 0.66998345719859 < 1e-8
    assert float(mn.value).hex() == NDVI_MIN_HEX and float(mx.value).hex() == NDVI_MAX_HEX


def test_raster_ingest_mirror(ec, golden_dir):
    """RasterBandEx mirror (src/gdal/rasterband.rs:82-125, src/gdal/mod.rs:49-70): read_cells / read_cells_masked,
    nodata f64 -> NoData<T> range check, row-block reads that tile the band."""
    from erased_cells_hip import raster, sharded
    rb = raster.RasterBand.open(os.path.join(golden_dir, "L8-Elkton-VA-B5-nd.tiff"))
    assert rb.size() == (186, 169) and rb.band_type() == ec.UInt16 and rb.no_data_value() == 0.0
    m = rb.read_cells_masked()
    assert m.counts() == (31430, 4) and m.cell_type() == ec.UInt16
    assert rb.read_cells().min_max() == (ec.CellValue(ec.UInt16, 0), ec.CellValue(ec.UInt16, 39368))
    nodata = 0
    for g in range(8):
        off, ln = sharded.shard_range(169, 186, g, 8)
        part = rb.read_cells_masked_rows(off // 186, ln // 186)
        assert part.buffer() == m.buffer().shard(off, ln)
        nodata += part.counts()[1]
    assert nodata == 4
    assert raster.nodata_from_f64(ec.UInt8, 255.9).value(ec.UInt8).value == 255
    assert raster.nodata_from_f64(ec.Float32, -9999.0).value(ec.Float32).value == np.float32(-9999.0)
    assert raster.nodata_from_f64(ec.Int16, None).value(ec.Int16) is None
    with pytest.raises(raster.NoDataConversionError):
        raster.nodata_from_f64(ec.UInt8, -9999.0)
    with pytest.raises(raster.NoDataConversionError):
        raster.nodata_from_f64(ec.Int16, float("nan"))


def test_ndvi_host_to_host_on_the_fixtures(ec, golden_dir):
    """The reference's two GDAL tests (src/gdal/rasterband.rs:138-191) with nothing resident on the GPU: the bands stay numpy
    arrays, NDVI is one streamed call (`ec_host_expr` / `ec_host_masked_expr`, 2000-cell chunks), the answers are the
    reference's (B.25-B.27)."""
    P = ec.fused
    red_c, red_nd = _band(golden_dir, "B4")
    nir_c, _ = _band(golden_dir, "B5")
    nd_c, nir_nd = _band(golden_dir, "B5-nd")
    S, R = (lambda k: k), (lambda k: 4 + k)
    ndvi = [(ec.SUB, S(0), S(1), 0), (ec.ADD, S(0), S(1), 1), (ec.DIV, R(0), R(1), 0)]
    out = P.program_host([nir_c.ravel(), red_c.ravel()], [], ndvi, chunk_cells=2000)
    assert float(out.min()).hex() == NDVI_MIN_HEX and float(out.max()).hex() == NDVI_MAX_HEX
    out, valid = P.program_host_masked([nd_c.ravel(), red_c.ravel()], [nir_nd, red_nd], [], ndvi, out_nodata=-9999.0, want_mask=True, chunk_cells=2000)
    assert (int(valid.sum()), int((~valid).sum())) == (31430, 4) and np.all(out[~valid] == -9999.0)
    assert float(out[valid].min()).hex() == NDVI_MIN_HEX and float(out[valid].max()).hex() == NDVI_MAX_HEX


def test_ndvi_fused_single_pass(ec, golden_dir):
    """The same two GDAL tests through the fused single-pass kernel: identical bits."""
    red_c, red_nd = _band(golden_dir, "B4")
    nir_c, nir_nd = _band(golden_dir, "B5-nd")
    red, nir = ec.CellBuffer.from_vec(red_c.ravel()), ec.CellBuffer.from_vec(_band(golden_dir, "B5")[0].ravel())
    ndvi = ec.fused.ndvi(nir, red)
    assert ndvi == (nir - red) / (nir + red)
    mn, mx = ndvi.min_max()
    assert float(mn.value).hex() == NDVI_MIN_HEX and float(mx.value).hex() == NDVI_MAX_HEX
    mred = ec.MaskedCellBuffer.from_vec_with_nodata(red_c.ravel(), ec.NoData.new(np.uint16(red_nd)))
    mnir = ec.MaskedCellBuffer.from_vec_with_nodata(nir_c.ravel(), ec.NoData.new(np.uint16(nir_nd)))
    m = ec.fused.ndvi(mnir, mred)
    assert m == (mnir - mred) / (mnir + mred)
    assert m.counts() == (31430, 4)
    mn, mx = m.min_max()
    assert float(mn.value).hex() == NDVI_MIN_HEX and float(mx.value).hex() == NDVI_MAX_HEX


def test_read_cells_masked_ndvi_sharded(ec, golden_dir):
    """src/gdal/rasterband.rs:166-191 (B.26, B.27) — and BASELINE config 5: convert u16->f32,
    NDVI, rows sharded 8 ways (22,21,...,21), shard results combined through the min/max keys."""
    from erased_cells_hip import sharded
    red_c, red_nd = _band(golden_dir, "B4")
    nir_c, nir_nd = _band(golden_dir, "B5-nd")
    rows, cols = red_c.shape
    red = ec.MaskedCellBuffer.from_vec_with_nodata(red_c.ravel(), ec.NoData.new(np.uint16(red_nd)))
    nir = ec.MaskedCellBuffer.from_vec_with_nodata(nir_c.ravel(), ec.NoData.new(np.uint16(nir_nd)))
    nir_data, nir_nodata = nir.counts()
    assert (nir_data, nir_nodata) == (31430, 4) and nir_data + nir_nodata == 186 * 169
    ndvi = (nir - red) / (nir + red)
    assert ndvi.counts() == (nir_data, nir_nodata)
    mn, mx = ndvi.min_max()
    assert float(mn.value).hex() == NDVI_MIN_HEX and float(mx.value).hex() == NDVI_MAX_HEX
    # 8 row-block shards on one GPU
    import ctypes as C
    key_lo, key_hi, data, nodata, lens = None, None, 0, 0, []
    for g in range(8):
        off, ln = sharded.shard_range(rows, cols, g, 8)
        lens.append(ln // cols)
        r32 = red.shard(off, ln).convert(ec.Float32)
        n32 = nir.shard(off, ln).convert(ec.Float32)
        part = (n32 - r32) / (n32 + r32)
        d, nd = part.counts()
        data, nodata = data + d, nodata + nd
        keys = ec.DeviceMem(16)
        ec._ffi.check(ec.lib().ec_min_max_keys(part.cell_type(), part.buffer().mem.ptr, part.mask().mem.ptr, part.len(),
                                               keys.ptr, None))
        k = np.empty(2, np.int64)
        ec._ffi.check(ec.lib().ec_download(k.ctypes.data_as(C.c_void_p), keys.ptr, 16, None))
        key_lo = int(k[0]) if key_lo is None else max(key_lo, int(k[0]))
        key_hi = int(k[1]) if key_hi is None else max(key_hi, int(k[1]))
    assert lens == [22, 21, 21, 21, 21, 21, 21, 21]
    assert (data, nodata) == (31430, 4)
    smn, smx = sharded.combine_min_max_keys(ec.Float64, (key_lo, key_hi))
    assert float(smn.value).hex() == NDVI_MIN_HEX and float(smx.value).hex() == NDVI_MAX_HEX


@pytest.mark.parametrize("world,fused", [(1, False), (2, False), (4, True)])
def test_config5_example_multi_process(world, fused):
    """examples/ndvi_sharded.py: one process per rank (gloo rehearsal of the RCCL exchange on one GPU)."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = os.path.join(root, "examples", "ndvi_sharded.py")
    extra = ["--backend", "gloo", "--single-device"] + (["--fused"] if fused else [])
    if world == 1:
        cmd = [sys.executable, script] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), script] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "matches the reference's known answers" in r.stdout
