"""GPU parity tests: the HIP path, called through the C ABI, against the oracle on
the same seeded inputs.  Bit-exact everywhere (integer, byte and index work, and
the f64 results too — north_star allows 1 ULP on the f64 path; we hold it to 0).

The only tolerance: a NaN whose sign/payload the reference itself does not pin —
both operands NaN in a commutative op, where the x86 result depends on which
operand the compiler put first — is compared by class.
"""
import os

import numpy as np
import pytest

from oracle import eco
from vectors import chain_loose, assert_f64_bits_equal, bits_of, rand_cells, rand_mask

pytestmark = pytest.mark.gpu

OPS = [eco.ADD, eco.SUB, eco.MUL, eco.DIV]
NT = eco.NTYPES


@pytest.fixture(scope="module")
def ec():
    import erased_cells_hip as ec
    ec.init(0)
    return ec


@pytest.fixture(params=[(1, 1, 256), (0, 1, 256), (1, 2, 256), (1, 0, 256), (1, 1, 0)],
                ids=["default", "cellwise-when-unaligned", "peel-2-byte-operands-too", "no-peel", "every-load-nt"])
def ua(ec, request):
    """Alternative paths; all must match the oracle.
    unaligned_vector: vector kernels with unaligned global access (default) or, knob off, the cell-wise kernels for
    windows that are not 16-byte aligned.
    peel: leading-cell peel of the binop/fused kernels for 1-byte operands (default), also 2-byte ones, or never.
    mall_mb: Infinity Cache budget of the launch's load policy — 256 (default: operands that fit are loaded with the
    default cache policy; test-sized operands all do) or 0 (every load non-temporal): the two arms of every policy branch."""
    ec.lib().ec_tune_set(b"unaligned_vector", request.param[0])
    ec.lib().ec_tune_set(b"peel", request.param[1])
    ec.lib().ec_tune_set(b"mall_mb", request.param[2])
    yield request.param
    ec.lib().ec_tune_set(b"unaligned_vector", 1)
    ec.lib().ec_tune_set(b"peel", 1)
    ec.lib().ec_tune_set(b"mall_mb", 256)


def _both_nan(l, r):
    with np.errstate(all="ignore"):
        return np.isnan(l.astype(np.float64)) & np.isnan(r.astype(np.float64))


# ---------------------------------------------------------------- binop: src/buffer.rs:324-329
@pytest.mark.parametrize("lct", range(NT))
def test_binop_all_pairs_bit_exact(ec, lct):
    n = 8 * 1024 + 37  # several wave tiles + ragged tail
    for rct in range(NT):
        l, r = rand_cells(lct, n, 11), rand_cells(rct, n, 12)
        dl, dr = ec.CellBuffer.from_vec(l), ec.CellBuffer.from_vec(r)
        for op in OPS:
            got = dl._binop(op, dr)
            assert got.cell_type() == ec.Float64 and got.len() == n
            exp = eco.f_binop(op, l, r)
            loose = _both_nan(l, r) if op in (eco.ADD, eco.MUL) else None
            assert_f64_bits_equal(got.to_numpy(), exp, nan_by_class_where=loose)


@pytest.mark.parametrize("variant", [-1, 0, 1])
def test_binop_reference_shaped_oracle_and_variants(ec, variant):
    """Both kernel variants (direct / LDS-staged; -1: the library's own choice) against the reference-shaped oracle."""
    ec.lib().ec_tune_set(b"binop_variant", variant)
    try:
        n = 3000
        for lct, rct in [(eco.U8, eco.U16), (eco.U16, eco.U8), (eco.I8, eco.F32), (eco.U64, eco.I64), (eco.F32, eco.F32),
                         (eco.U8, eco.U8), (eco.I16, eco.U32), (eco.F64, eco.U16)]:
            l, r = rand_cells(lct, n, 21), rand_cells(rct, n, 22)
            for op in OPS:
                got = ec.CellBuffer.from_vec(l)._binop(op, ec.CellBuffer.from_vec(r)).to_numpy()
                loose = _both_nan(l, r) if op in (eco.ADD, eco.MUL) else None
                assert_f64_bits_equal(got, eco.binop(op, l, r), nan_by_class_where=loose)
    finally:
        ec.lib().ec_tune_set(b"binop_variant", -1)  # the default: by rule


@pytest.mark.parametrize("variant", [0, 1])
def test_binop_lengths_tails_and_alignment(ec, ua, variant):
    """Ragged sizes around every tile boundary, zip truncation, and odd cell offsets
    (row-block windows that are not 16-byte aligned: fixture `ua`)."""
    ec.lib().ec_tune_set(b"binop_variant", variant)
    try:
        big = 2 * 1024 * 1024 + 5
        l, r = eco.fill_u8(big, 0x5EED0001), eco.fill_u16(big, 0x5EED0002, lo=0)  # zeros in the divisor: 0/0, x/0
        dl, dr = ec.CellBuffer.from_vec(l), ec.CellBuffer.from_vec(r)
        for n in [1, 2, 3, 63, 64, 127, 128, 129, 255, 256, 1023, 1024, 1025, 2047, 2048, 4095, 4096, 4097, 8191, 8193,
                  65536 + 1, 1024 * 1024 - 1, big]:
            got = (dl.shard(0, n) / dr.shard(0, n)).to_numpy()
            assert_f64_bits_equal(got, eco.f_binop(eco.DIV, l[:n], r[:n]))
        for off_l, off_r, n in [(1, 0, 5000), (0, 1, 5000), (3, 5, 4097), (8, 8, 9000), (16, 8, 70000), (16, 32, 70001)]:
            got = (dl.shard(off_l, n) / dr.shard(off_r, n)).to_numpy()
            assert_f64_bits_equal(got, eco.f_binop(eco.DIV, l[off_l:off_l + n], r[off_r:off_r + n]))
        # zip: result length is the shorter operand's (buffer.rs:327)
        got = dl.shard(0, 700) - dr.shard(0, 333)
        assert got.len() == 333
        assert_f64_bits_equal(got.to_numpy(), eco.f_binop(eco.SUB, l[:700], r[:333]))
        # empty result is a UInt8 buffer (buffer.rs:233-234)
        e = dl.shard(0, 0) + dr
        assert e.cell_type() == ec.UInt8 and e.len() == 0
    finally:
        ec.lib().ec_tune_set(b"binop_variant", -1)  # the default: by rule


def test_binop_lds_staged_variant_by_rule(ec):
    """An 8-byte operand against one of <= 4 bytes, a million cells or more, nothing kept cacheable: the library takes the LDS-staged
    kernel under an occupancy cap by itself (ec_binop_tu.hpp).  Same cells as the oracle, ragged tail included, and the counter says
    the rule fired; below the size threshold, with the knob at 0, and for pairs outside the rule it does not."""
    L = ec.lib()
    import ctypes as C
    def launches():
        v = C.c_int64()
        ec._ffi.check(L.ec_stat_get(b"binop_lds_rule_launches", C.byref(v)))
        return v.value
    n = (1 << 20) + 777
    pairs = [(eco.F64, eco.U16), (eco.U16, eco.F64), (eco.F64, eco.F32), (eco.F32, eco.F64), (eco.F64, eco.U8), (eco.I8, eco.F64),
             (eco.I64, eco.U32), (eco.I16, eco.U64)]
    # operands this small would be kept cacheable (they fit the Infinity Cache) and the rule leaves those launches to the direct kernel:
    # a cache budget of 0 gives these megabyte-sized buffers the load policy of the gigabyte-sized ones the rule is for
    mall = C.c_int64()
    ec._ffi.check(L.ec_stat_get(b"tune.mall_mb", C.byref(mall)))
    L.ec_tune_set(b"mall_mb", 0)
    try:
        _lds_rule_cases(ec, L, launches, n, pairs)
    finally:
        L.ec_tune_set(b"mall_mb", mall.value)
    l, r = rand_cells(eco.F64, n, 35), rand_cells(eco.U16, n, 36)
    before = launches()
    got = (ec.CellBuffer.from_vec(l) * ec.CellBuffer.from_vec(r)).to_numpy()  # cacheable operands: the direct kernel
    assert launches() == before
    assert_f64_bits_equal(got, eco.f_binop(eco.MUL, l, r), nan_by_class_where=_both_nan(l, r))


def _lds_rule_cases(ec, L, launches, n, pairs):
    for lct, rct in pairs:
        l, r = rand_cells(lct, n, 31), rand_cells(rct, n, 32)
        dl, dr = ec.CellBuffer.from_vec(l), ec.CellBuffer.from_vec(r)
        for op in OPS:
            before = launches()
            got = dl._binop(op, dr).to_numpy()
            assert launches() == before + 1, (lct, rct, op)
            loose = _both_nan(l, r) if op in (eco.ADD, eco.MUL) else None
            assert_f64_bits_equal(got, eco.f_binop(op, l, r), nan_by_class_where=loose)
        before = launches()
        small = dl.shard(0, (1 << 20) - 1) + dr.shard(0, (1 << 20) - 1)  # below the threshold: the direct kernel
        assert launches() == before and small.len() == (1 << 20) - 1
        L.ec_tune_set(b"binop_variant", 0)
        try:
            got0 = (dl / dr).to_numpy()
        finally:
            L.ec_tune_set(b"binop_variant", -1)
        assert launches() == before
        assert_f64_bits_equal(got0, eco.f_binop(eco.DIV, l, r))
    for lct, rct in [(eco.F64, eco.F64), (eco.U8, eco.U16), (eco.F32, eco.F32), (eco.U16, eco.U32)]:  # outside the rule
        l, r = rand_cells(lct, n, 33), rand_cells(rct, n, 34)
        before = launches()
        got = (ec.CellBuffer.from_vec(l) - ec.CellBuffer.from_vec(r)).to_numpy()
        assert launches() == before
        assert_f64_bits_equal(got, eco.f_binop(eco.SUB, l, r))


def test_div_by_zero_and_nan_policy(ec):
    """int ÷ 0 -> ±inf, 0 ÷ 0 -> the x86 default NaN 0xFFF8…, operand NaNs propagate lhs-first, quieted."""
    a = ec.CellBuffer.from_vec(np.array([0, 1, 7, 0], np.int8)) / ec.CellBuffer.from_vec(np.array([0, 0, 0, 5], np.uint16))
    assert bits_of(a.to_numpy()).tolist() == [0xFFF8000000000000, 0x7FF0000000000000, 0x7FF0000000000000, 0]
    a = ec.CellBuffer.from_vec(np.array([-3], np.int32)) / ec.CellBuffer.from_vec(np.array([0], np.uint8))
    assert bits_of(a.to_numpy()).tolist() == [0xFFF0000000000000]
    snan = np.array([0x7FF0000000000001, 0xFFF4000000000002], np.uint64).view(np.float64)
    one = np.ones(2)
    for op in OPS:
        got = ec.CellBuffer.from_vec(snan)._binop(op, ec.CellBuffer.from_vec(one)).to_numpy()
        assert bits_of(got).tolist() == [0x7FF8000000000001, 0xFFFC000000000002]
        got = ec.CellBuffer.from_vec(one)._binop(op, ec.CellBuffer.from_vec(snan)).to_numpy()
        assert bits_of(got).tolist() == [0x7FF8000000000001, 0xFFFC000000000002]
    inf = np.array([np.inf])
    assert bits_of((ec.CellBuffer.from_vec(inf) - ec.CellBuffer.from_vec(inf)).to_numpy())[0] == 0xFFF8000000000000
    assert bits_of((ec.CellBuffer.from_vec(inf) * ec.CellBuffer.from_vec(np.zeros(1))).to_numpy())[0] == 0xFFF8000000000000
    # f32 NaN payloads widen like cvtss2sd (payload << 29, quieted)
    f = np.array([0x7FC00001, 0xFF800001], np.uint32).view(np.float32)
    got = (ec.CellBuffer.from_vec(f) + ec.CellBuffer.from_vec(np.zeros(2, np.uint8))).to_numpy()
    assert_f64_bits_equal(got, eco.f_binop(eco.ADD, f, np.zeros(2, np.uint8)))


# ---------------------------------------------------------------- scalar rhs: src/buffer.rs:346-352
@pytest.mark.parametrize("lct", range(NT))
def test_binop_scalar_bit_exact(ec, lct):
    n = 5001
    l = rand_cells(lct, n, 31)
    dl = ec.CellBuffer.from_vec(l)
    with np.errstate(all="ignore"):
        lnan = np.isnan(l.astype(np.float64))
    for sct in range(NT):
        for sval in (rand_cells(sct, 3, 32, specials=False)[1], 0, 2):
            s_o = eco.Value.of(sct, sval)
            s_d = ec.CellValue(sct, sval)
            for op in OPS:
                got = dl._binop(op, s_d)
                assert got.cell_type() == ec.Float64
                assert_f64_bits_equal(got.to_numpy(), eco.f_binop_scalar(op, l, s_o))
    nan_s = ec.CellValue(ec.Float64, np.array([0xFFF0000000000123], np.uint64).view(np.float64)[0])
    got = (dl * nan_s).to_numpy()
    exp = eco.f_binop_scalar(eco.MUL, l, eco.Value.from_bits(eco.F64, 0xFFF0000000000123))
    assert_f64_bits_equal(got, exp, nan_by_class_where=lnan)
    assert (dl.shard(0, 0) * 2.0).cell_type() == ec.UInt8


# ---------------------------------------------------------------- neg: src/buffer.rs:360-365
@pytest.mark.parametrize("ct", range(NT))
def test_neg_bit_exact(ec, ua, ct):
    for n in (1, 15, 16, 17, 4099, 70001):
        a = rand_cells(ct, n, 41)
        got = -ec.CellBuffer.from_vec(a)
        exp = eco.neg(a) if n < 5000 else eco.f_neg(a)
        assert ec.NP_DTYPES[got.cell_type()] == exp.dtype
        assert np.array_equal(bits_of(got.to_numpy()), bits_of(exp))
    got = -(ec.CellBuffer.from_vec(rand_cells(ct, 100, 42)).shard(3, 50))  # unaligned window
    assert np.array_equal(bits_of(got.to_numpy()), bits_of(eco.f_neg(rand_cells(ct, 100, 42)[3:53])))
    assert (-ec.CellBuffer.empty(0, ct)).cell_type() == ec.UInt8


# ---------------------------------------------------------------- convert: src/buffer.rs:150-167
@pytest.mark.parametrize("sct", range(NT))
def test_convert_all_pairs(ec, sct):
    n = 9001
    a = rand_cells(sct, n, 51)
    da = ec.CellBuffer.from_vec(a)
    for dct in range(NT):
        if eco.can_fit_into(sct, dct):
            got = da.convert(dct)
            assert got.cell_type() == dct and got.len() == n
            assert np.array_equal(bits_of(got.to_numpy()), bits_of(eco.f_convert(a, dct)))
            assert np.array_equal(bits_of(da.shard(1, 777).convert(dct).to_numpy()), bits_of(eco.convert(a[1:778], dct)))
            assert np.array_equal(bits_of(da.to_vec(dct)), bits_of(eco.f_convert(a, dct)))
        else:
            with pytest.raises(ec.NarrowingError) as ei:
                da.convert(dct)
            assert (ei.value.src, ei.value.dst) == (sct, dct)
            assert "Invalid narrowing from cell-type" in str(ei.value)
    # empty convert to another type collects nothing -> UInt8 (buffer.rs:233-234)
    if sct != eco.F64:
        assert ec.CellBuffer.empty(0, sct).convert(eco.F64).cell_type() == ec.UInt8


# ---------------------------------------------------------------- min_max: src/buffer.rs:169-173, masked_buffer.rs:208-217
@pytest.mark.parametrize("ct", range(NT))
def test_min_max_total_order(ec, ua, ct):
    for n in (0, 1, 5, 255, 4096, 100003, 1 << 21):
        a = rand_cells(ct, n, 61)
        d = ec.CellBuffer.from_vec(a)
        for mask in (None, rand_mask(n, 62), np.zeros(n, np.uint8)):
            if mask is None:
                mn, mx = d.min_max()
            else:
                mn, mx = ec.MaskedCellBuffer(d, ec.Mask.new(mask)).min_max()
            emn, emx = eco.f_min_max(a, mask)
            assert (mn.ct, mn.bits(), mx.ct, mx.bits()) == (emn.ct, emn.bits(), emx.ct, emx.bits()), (n, mask is None)
    a = rand_cells(ct, 5000, 63)
    mn, mx = ec.CellBuffer.from_vec(a).shard(1, 4001).min_max()  # unaligned window
    emn, emx = eco.min_max(a[1:4002])
    assert (mn.bits(), mx.bits()) == (emn.bits(), emx.bits())


# ---------------------------------------------------------------- masks: src/masked/*.rs
@pytest.mark.parametrize("ct", range(NT))
def test_mask_from_nodata_and_select(ec, ct):
    n = 10007
    a = rand_cells(ct, n, 71)
    a[::7] = a[3]  # make the chosen nodata value frequent
    d = ec.CellBuffer.from_vec(a)
    for nd_o, nd_d in ((eco.nodata_value(eco.ND_DEFAULT, ct), ec.NoData.default()),
                       (eco.Value.of(ct, a[3]), ec.NoData.new(ec.CellValue(ct, a[3]))),
                       (None, ec.NoData.none())):
        m = ec.mask_from_nodata(d, nd_d)
        assert np.array_equal(m.to_numpy(), eco.f_mask_from_nodata(a, nd_o))
        mk = rand_mask(n, 72)
        got = ec.MaskedCellBuffer(d, ec.Mask.new(mk)).to_vec_with_nodata(ct, nd_d)
        assert np.array_equal(bits_of(got), bits_of(eco.f_mask_select(a, mk, nd_o)))
    sub = ec.mask_from_nodata(d.shard(5, 333), ec.NoData.default())
    assert np.array_equal(sub.to_numpy(), eco.mask_from_nodata(a[5:338], eco.ND_DEFAULT))


def test_float_nodata_is_bitwise(ec):
    """Only the NaN with identical bits matches; -0.0 != +0.0 (nodata.rs:42-49 -> value.rs:248-271)."""
    nan_c = np.array([0x7FF8000000000000], np.uint64).view(np.float64)[0]
    v = np.array([1.0, nan_c, -nan_c, 0.0, -0.0])
    v[2] = np.array([0xFFF8000000000000], np.uint64).view(np.float64)[0]
    m = ec.MaskedCellBuffer.from_vec_with_nodata(v, ec.NoData.default()).mask().to_numpy()
    assert m.tolist() == [1, 0, 1, 1, 1]
    m = ec.MaskedCellBuffer.from_vec_with_nodata(v, ec.NoData.new(0.0)).mask().to_numpy()
    assert m.tolist() == [1, 1, 1, 0, 1]


def test_mask_logic_and_counts(ec, ua):
    for n in (0, 1, 15, 16, 17, 4095, 4097, 1 << 20, (1 << 22) + 3):
        a, b = rand_mask(n, 81), rand_mask(n, 82, 0.4)
        da, db = ec.Mask.new(a), ec.Mask.new(b)
        assert np.array_equal((da & db).to_numpy(), eco.mask_and(a, b))
        assert np.array_equal((da | db).to_numpy(), eco.mask_or(a, b))
        assert np.array_equal((~da).to_numpy(), eco.mask_not(a))
        assert da.counts() == eco.mask_counts(a)
        assert da.all(True) == eco.mask_all(a, True) and da.all(False) == eco.mask_all(a, False)
    a, b = rand_mask(1000, 83), rand_mask(600, 84)
    assert (ec.Mask.new(a) & ec.Mask.new(b)).len() == 600  # borrowed forms zip (mask.rs:129-140)
    m = ec.Mask.new(a)
    m &= ec.Mask.new(b)                                     # owned form keeps lhs length (mask.rs:118-127)
    assert m.len() == 1000
    assert np.array_equal(m.to_numpy(), np.concatenate([a[:600] & b, a[600:]]))
    assert ec.Mask.new(a).shard(7, 500).counts() == eco.mask_counts(a[7:507])  # unaligned window
    # in-place (owned) forms on ragged windows at odd offsets: out aliases lhs (mask.rs:103-109,118-127,142-151)
    import ctypes as C
    L = ec.lib()
    for n, off in ((70001, 3), (4099, 1), (33, 13)):
        a, b = rand_mask(n + off, 85), rand_mask(n + off, 86, 0.5)
        for name, fn in (("and", eco.mask_and), ("or", eco.mask_or)):
            whole, rhs = ec.Mask.new(a), ec.Mask.new(b)
            w, r = whole.shard(off, n), rhs.shard(off, n)
            if name == "and":
                w &= r
            else:
                w |= r
            assert np.array_equal(whole.to_numpy(), np.concatenate([a[:off], fn(a[off:], b[off:])])), (name, n, off)
        whole = ec.Mask.new(a)
        w = whole.shard(off, n)
        ec._ffi.check(L.ec_mask_not(w.mem.ptr, n, w.mem.ptr, ec.stream()))  # Not for Mask, in place
        assert np.array_equal(whole.to_numpy(), np.concatenate([a[:off], eco.mask_not(a[off:])]))


# ---------------------------------------------------------------- masked binop: src/masked/masked_buffer.rs:326-335
@pytest.mark.parametrize("variant", [0, 1])
def test_masked_binop_fused(ec, ua, variant):
    ec.lib().ec_tune_set(b"binop_variant", variant)
    try:
        for lct, rct in [(eco.F32, eco.F32), (eco.U8, eco.U16), (eco.F64, eco.F32), (eco.I64, eco.U8)]:
            for n in (1, 17, 4096, 50001):
                l, r = rand_cells(lct, n, 91), rand_cells(rct, n, 92)
                lm, rm = rand_mask(n, 93), rand_mask(n, 94)
                ml = ec.MaskedCellBuffer(ec.CellBuffer.from_vec(l), ec.Mask.new(lm))
                mr = ec.MaskedCellBuffer(ec.CellBuffer.from_vec(r), ec.Mask.new(rm))
                for op in OPS:
                    got = ml._binop(op, mr)
                    loose = _both_nan(l, r) if op in (eco.ADD, eco.MUL) else None
                    # masked-out cells are still computed (masked_buffer.rs:331)
                    assert_f64_bits_equal(got.buffer().to_numpy(), eco.f_binop(op, l, r), nan_by_class_where=loose)
                    assert np.array_equal(got.mask().to_numpy(), eco.mask_and(lm, rm))
        # unaligned windows (fixture `ua`: vector kernel or cell-wise kernel)
        l, r = rand_cells(eco.F32, 3000, 95), rand_cells(eco.F32, 3000, 96)
        lm, rm = rand_mask(3000, 97), rand_mask(3000, 98)
        ml = ec.MaskedCellBuffer(ec.CellBuffer.from_vec(l), ec.Mask.new(lm)).shard(1, 2000)
        mr = ec.MaskedCellBuffer(ec.CellBuffer.from_vec(r), ec.Mask.new(rm)).shard(3, 2000)
        got = ml + mr
        assert_f64_bits_equal(got.buffer().to_numpy(), eco.f_binop(eco.ADD, l[1:2001], r[3:2003]),
                              nan_by_class_where=_both_nan(l[1:2001], r[3:2003]))
        assert np.array_equal(got.mask().to_numpy(), lm[1:2001] & rm[3:2003])
    finally:
        ec.lib().ec_tune_set(b"binop_variant", -1)  # the default: by rule


# ---------------------------------------------------------------- Ord / Eq on the device: src/buffer.rs:373-436
@pytest.mark.parametrize("ct", range(NT))
def test_buffer_cmp_on_device(ec, ua, ct):
    import ctypes as C
    n = 300007
    a = rand_cells(ct, n, 101)
    da = ec.CellBuffer.from_vec(a)
    assert da == da and da.cmp(ec.CellBuffer.from_vec(a.copy())) == 0  # NaNs equal themselves bitwise (buffer.rs:624-626)
    rng = np.random.default_rng(5)
    for pos in [0, 1, 15, 16, 4095, 4096, n // 2 + 3, n - 1]:
        b = a.copy()
        repl = rand_cells(ct, 64, 102 + pos)
        repl = repl[bits_of(repl) != bits_of(a[pos:pos + 1])[0]]
        b[pos] = repl[0]
        if pos + 1000 < n:
            b[pos + 1000] = repl[-1]  # a later difference must not matter
        db = ec.CellBuffer.from_vec(b)
        assert da.cmp(db) == eco.buffer_cmp(a, b), pos
        assert db.cmp(da) == eco.buffer_cmp(b, a), pos
        idx = C.c_uint64()
        ec._ffi.check(ec.lib().ec_first_difference(ct, da.mem.ptr, db.mem.ptr, n, C.byref(idx), None))
        assert idx.value == pos
        assert (da == db) is False
    # prefix relation: shorter sorts first; unaligned windows; cell type decides before content
    assert da.shard(0, 1000).cmp(da.shard(0, 999)) == 1 and da.shard(0, 999).cmp(da.shard(0, 1000)) == -1
    assert da.shard(1, 5000).cmp(ec.CellBuffer.from_vec(a[1:5001])) == 0
    assert da.shard(3, 5000).cmp(ec.CellBuffer.from_vec(a[4:5004])) == eco.buffer_cmp(a[3:5003], a[4:5004])
    other = ec.CellBuffer.with_defaults(5, (ct + 1) % NT)
    assert da.cmp(other) == (-1 if ct < (ct + 1) % NT else 1)
    assert ec.CellBuffer.empty(0, ct).cmp(ec.CellBuffer.empty(0, ct)) == 0


# ---------------------------------------------------------------- fused chains == eager chains
def test_fused_expression_equals_eager_chain(ec, ua):
    """(x o1 y) o2 (z o3 w) in one pass is bit-identical to the reference's eager operator chain."""
    rng = np.random.default_rng(77)
    for trial in range(40):
        cts = [int(c) for c in rng.integers(0, NT, size=4)]
        o1, o2, o3 = (int(o) for o in rng.integers(0, 4, size=3))
        n = int(rng.choice([1, 2, 3, 511, 512, 513, 1024, 4097, 20001]))
        h = [rand_cells(ct, n, 200 + trial * 4 + k) for k, ct in enumerate(cts)]
        d = [ec.CellBuffer.from_vec(a) for a in h]
        four = trial % 2 == 0
        if trial % 5 == 0:  # aliased operands, NDVI-shaped
            h[2], h[3], d[2], d[3] = h[0], h[1], d[0], d[1]
        got = ec.fused.expr(d[0], o1, d[1], o2, d[2], o3 if four else ec.fused.OP_NONE, d[3] if four else None)
        t1 = d[0]._binop(o1, d[1])
        t2 = d[2]._binop(o3, d[3]) if four else d[2]
        exp = t1._binop(o2, t2)
        assert got.cell_type() == ec.Float64 and got.len() == n
        assert np.array_equal(bits_of(got.to_numpy()), bits_of(exp.to_numpy())), (trial, cts, (o1, o2, o3), n)
        # and against the oracle, under the single-op rule carried through the chain: NaNs bit for bit, except where
        # both operands of a commutative step are NaN (or an input of a step was such a cell)
        e1 = eco.f_binop(o1, h[0], h[1])
        e2 = eco.f_binop(o3, h[2], h[3]) if four else h[2]
        eo = eco.f_binop(o2, e1, e2)
        loose = chain_loose(o1, h[0], h[1], o2, e1, e2, o3 if four else None, h[2] if four else None, h[3] if four else None)
        assert_f64_bits_equal(got.to_numpy(), eo, nan_by_class_where=loose)
    # unaligned windows (fixture `ua`)
    a, b = rand_cells(eco.U16, 5000, 301), rand_cells(eco.U16, 5000, 302)
    da, db = ec.CellBuffer.from_vec(a).shard(1, 4000), ec.CellBuffer.from_vec(b).shard(3, 4000)
    got = ec.fused.ndvi(da, db)
    exp = (da - db) / (da + db)
    assert np.array_equal(bits_of(got.to_numpy()), bits_of(exp.to_numpy()))
    assert ec.fused.ndvi(da.shard(0, 0), db).cell_type() == ec.UInt8  # empty chain -> UInt8 (buffer.rs:233-234)
    # zip truncation of every step
    assert ec.fused.add_mul(da, db.shard(0, 100), da).len() == 100
    # scalar operands (the RHS-scalar operator form, src/buffer.rs:346-352), any position, any scalar type
    for n in (1, 513, 20001):
        x, y = rand_cells(eco.I16, n, 310), rand_cells(eco.F32, n, 311)
        dx, dy = ec.CellBuffer.from_vec(x), ec.CellBuffer.from_vec(y)
        got = ec.fused.expr(dx, ec.ADD, dy, ec.MUL, 2.0)                       # (x + y) * 2.0
        assert np.array_equal(bits_of(got.to_numpy()), bits_of(((dx + dy) * 2.0).to_numpy()))
        got = ec.fused.expr(dx, ec.MUL, np.float32(0.0001), ec.ADD, dy)        # x * scale + y
        assert np.array_equal(bits_of(got.to_numpy()), bits_of(((dx * np.float32(0.0001)) + dy).to_numpy()))
        got = ec.fused.expr(dx, ec.SUB, 7, ec.DIV, dx, ec.ADD, np.uint8(3))    # (x - 7) / (x + 3), aliased + scalars
        assert np.array_equal(bits_of(got.to_numpy()), bits_of(((dx - 7) / (dx + np.uint8(3))).to_numpy()))
        got = ec.fused.expr(dx, ec.DIV, dx, ec.MUL, dx, ec.SUB, 1.5)           # same-type fast path with a scalar
        assert np.array_equal(bits_of(got.to_numpy()), bits_of(((dx / dx) * (dx - 1.5)).to_numpy()))


def test_fused_masked_expression_equals_eager_chain(ec):
    for n in (1, 17, 4096, 50001):
        cells = [rand_cells(ct, n, 400 + k) for k, ct in enumerate((eco.F32, eco.F32, eco.F32, eco.U16))]
        masks = [rand_mask(n, 410 + k) for k in range(4)]
        m = [ec.MaskedCellBuffer(ec.CellBuffer.from_vec(c), ec.Mask.new(k)) for c, k in zip(cells, masks)]
        got = ec.fused.add_mul(m[0], m[1], m[2])  # BASELINE config 3
        exp = (m[0] + m[1]) * m[2]
        assert np.array_equal(bits_of(got.buffer().to_numpy()), bits_of(exp.buffer().to_numpy()))
        assert np.array_equal(got.mask().to_numpy(), masks[0] & masks[1] & masks[2])
        got = ec.fused.expr(m[0], ec.SUB, m[3], ec.DIV, m[1], ec.MUL, m[2])
        exp = (m[0] - m[3]) / (m[1] * m[2])
        assert np.array_equal(bits_of(got.buffer().to_numpy()), bits_of(exp.buffer().to_numpy()))
        assert np.array_equal(got.mask().to_numpy(), masks[0] & masks[1] & masks[2] & masks[3])
        got = ec.fused.ndvi(m[3], m[0])  # aliased operands: each mask ANDed once
        exp = (m[3] - m[0]) / (m[3] + m[0])
        assert np.array_equal(bits_of(got.buffer().to_numpy()), bits_of(exp.buffer().to_numpy()))
        assert np.array_equal(got.mask().to_numpy(), exp.mask().to_numpy())
        assert got.counts() == exp.counts()


def test_randomised_shapes_types_and_windows(ec, ua):
    """Property test (hypothesis): any type pair, op, length and pair of window offsets gives the oracle's
    bits — exercises every vector/cell-wise path choice, tile boundary and ragged tail."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    pool = {ct: rand_cells(ct, 70000, 900 + ct) for ct in range(NT)}
    dev = {ct: ec.CellBuffer.from_vec(a) for ct, a in pool.items()}
    mpool = rand_mask(70000, 950)
    dmask = ec.Mask.new(mpool)

    @settings(max_examples=int(os.environ.get("EC_PROP_EXAMPLES", "300")), deadline=None, suppress_health_check=list(HealthCheck))
    @given(lt=st.integers(0, NT - 1), rt=st.integers(0, NT - 1), op=st.integers(0, 3),
           n=st.one_of(st.integers(0, 40), st.integers(500, 1100), st.integers(4000, 4200), st.integers(60000, 65000)),
           lo=st.sampled_from([0, 1, 2, 3, 5, 8, 16, 17, 32, 48, 64, 1000]),
           ro=st.sampled_from([0, 1, 4, 7, 16, 31, 32, 64, 999]),
           variant=st.integers(0, 1), kind=st.integers(0, 8))
    def run(lt, rt, op, n, lo, ro, variant, kind):
        l, r = pool[lt][lo:lo + n], pool[rt][ro:ro + n]
        dl, dr = dev[lt].shard(lo, n), dev[rt].shard(ro, n)
        ec.lib().ec_tune_set(b"binop_variant", variant)
        if kind == 0:      # buffer op buffer
            got = dl._binop(op, dr)
            if n == 0:
                assert got.cell_type() == ec.UInt8 and got.len() == 0
                return
            loose = _both_nan(l, r) if op in (eco.ADD, eco.MUL) else None
            assert_f64_bits_equal(got.to_numpy(), eco.f_binop(op, l, r), nan_by_class_where=loose)
        elif kind == 1:    # masked op masked (fused value + mask kernel)
            ml = ec.MaskedCellBuffer(dl, dmask.shard(lo, n))
            mr = ec.MaskedCellBuffer(dr, dmask.shard(ro, n))
            got = ml._binop(op, mr)
            if n:
                loose = _both_nan(l, r) if op in (eco.ADD, eco.MUL) else None
                assert_f64_bits_equal(got.buffer().to_numpy(), eco.f_binop(op, l, r), nan_by_class_where=loose)
                assert np.array_equal(got.mask().to_numpy(), mpool[lo:lo + n] & mpool[ro:ro + n])
        elif kind == 2:    # neg + min_max (masked)
            if n:
                assert np.array_equal(bits_of((-dl).to_numpy()), bits_of(eco.f_neg(l)))
            mn, mx = ec.MaskedCellBuffer(dl, dmask.shard(ro, n)).min_max()
            emn, emx = eco.f_min_max(l, mpool[ro:ro + n])
            assert (mn.bits(), mx.bits()) == (emn.bits(), emx.bits())
        elif kind == 3:    # convert (legal pairs) / narrowing error
            if eco.can_fit_into(lt, rt):
                got = dl.convert(rt)
                if n and lt != rt:
                    assert np.array_equal(bits_of(got.to_numpy()), bits_of(eco.f_convert(l, rt)))
            else:
                with pytest.raises(ec.NarrowingError):
                    dl.convert(rt)
        elif kind == 4:    # compare + mask from nodata
            assert dl.cmp(dr) == eco.buffer_cmp(l, r)
            nd = eco.nodata_value(eco.ND_DEFAULT, lt)
            assert np.array_equal(ec.mask_from_nodata(dl, ec.NoData.default()).to_numpy(), eco.f_mask_from_nodata(l, nd))
        elif kind == 5:    # buffer op scalar (the scalar: a finite cell of the other pool)
            sval = pool[rt][ro + 70000 - 1000]
            with np.errstate(all="ignore"):
                if np.isnan(np.float64(sval)):
                    sval = pool[rt].dtype.type(3)
            got = dl._binop(op, ec.CellValue(rt, sval))
            if n == 0:
                assert got.cell_type() == ec.UInt8 and got.len() == 0
            else:
                assert_f64_bits_equal(got.to_numpy(), eco.f_binop_scalar(op, l, eco.Value.of(rt, sval)))
        elif kind == 6:    # mask logic, counts and select on windows
            ma, mb = dmask.shard(lo, n), dmask.shard(ro, n)
            ha, hb = mpool[lo:lo + n], mpool[ro:ro + n]
            assert np.array_equal((ma & mb).to_numpy(), eco.mask_and(ha, hb))
            assert np.array_equal((ma | mb).to_numpy(), eco.mask_or(ha, hb))
            assert np.array_equal((~ma).to_numpy(), eco.mask_not(ha))
            assert ma.counts() == eco.mask_counts(ha)
            nd = eco.nodata_value(eco.ND_DEFAULT, lt)
            got = ec.MaskedCellBuffer(dl, mb).to_vec_with_nodata(lt, ec.NoData.default())
            assert np.array_equal(bits_of(got), bits_of(eco.f_mask_select(l, hb, nd)))
        else:              # fused chains on windows == the eager chain (kind 7: four operands, kind 8: scalar third)
            o2, o3 = (op + 1) % 4, (op + 2) % 4
            if kind == 7:
                got = ec.fused.expr(dl, op, dr, o2, dr, o3, dl)
                exp = (dl._binop(op, dr))._binop(o2, dr._binop(o3, dl))
            else:
                got = ec.fused.expr(dl, op, dr, o2, 2.5)
                exp = (dl._binop(op, dr))._binop(o2, 2.5)
            assert got.cell_type() == exp.cell_type() and got.len() == exp.len()
            if n:
                assert np.array_equal(bits_of(got.to_numpy()), bits_of(exp.to_numpy()))

    try:
        run()
    finally:
        ec.lib().ec_tune_set(b"binop_variant", -1)  # the default: by rule


@pytest.mark.parametrize("map_u,reduce_bpc,reduce_shape", [(1, 1, 0), (4, 16, 1), (2, 3, 2), (2, 0, 3), (2, 0, 4)])
def test_tuning_knobs_do_not_change_results(ec, map_u, reduce_bpc, reduce_shape):
    """The non-default launch shapes behind ec_tune_set("map_u" / "reduce_bpc") are separate kernel instantiations
    and grid sizes; every map kernel and reduction must give the oracle's bits with them too."""
    L = ec.lib()
    L.ec_tune_set(b"map_u", map_u)
    L.ec_tune_set(b"reduce_bpc", reduce_bpc)
    L.ec_tune_set(b"reduce_shape", reduce_shape)
    try:
        n = 300001
        m1, m2 = rand_mask(n, 71), rand_mask(n, 72)
        dm1, dm2 = ec.Mask.new(m1), ec.Mask.new(m2)
        assert np.array_equal((dm1 & dm2).to_numpy(), eco.mask_and(m1, m2))
        assert np.array_equal((dm1 | dm2).to_numpy(), eco.mask_or(m1, m2))
        assert np.array_equal((~dm1).to_numpy(), eco.mask_not(m1))
        assert dm1.counts() == eco.mask_counts(m1)
        for ct in range(NT):
            a = rand_cells(ct, n, 700 + ct)
            d = ec.CellBuffer.from_vec(a)
            assert np.array_equal(bits_of((-d).to_numpy()), bits_of(eco.f_neg(a)))
            for dst in (eco.F64, eco.union(ct, eco.I16)):
                if eco.can_fit_into(ct, dst) and dst != ct:
                    assert np.array_equal(bits_of(d.convert(dst).to_numpy()), bits_of(eco.f_convert(a, dst)))
            for mask, dmask in ((None, None), (m1, dm1)):
                got = (d.min_max() if mask is None else ec.MaskedCellBuffer(d, dmask).min_max())
                exp = eco.f_min_max(a, mask)
                assert (got[0].bits(), got[1].bits()) == (exp[0].bits(), exp[1].bits())
            nd = eco.nodata_value(eco.ND_DEFAULT, ct)
            assert np.array_equal(ec.mask_from_nodata(d, ec.NoData.default()).to_numpy(), eco.f_mask_from_nodata(a, nd))
            sel = ec.MaskedCellBuffer(d, dm2).to_vec_with_nodata(ct, ec.NoData.default())
            assert np.array_equal(bits_of(sel), bits_of(eco.f_mask_select(a, m2, nd)))
            b = a.copy()
            b[n - 7] = a[0]
            assert d.cmp(ec.CellBuffer.from_vec(b)) == eco.buffer_cmp(a, b)
            f = ec.CellBuffer.fill(n, ec.CellValue(ct, a[3]))
            assert np.array_equal(bits_of(f.to_numpy()), bits_of(np.full(n, a[3], dtype=a.dtype)))
    finally:
        L.ec_tune_set(b"map_u", 2)
        L.ec_tune_set(b"reduce_bpc", 0)
        L.ec_tune_set(b"reduce_shape", 0)


def test_mask_counts_in_one_launch_equals_two_launches_and_the_oracle(ec):
    """`Mask::counts` (src/masked/mask.rs:72-80) in ONE launch — every workgroup adds (1 << 40 | its count) to a word of the stream's
    scratch, the last one writes the result and zeroes the word — against the partials + finalize form (`counts_one_launch` = 0) and the
    oracle: sizes around the one-workgroup boundary and far beyond it, odd window offsets, call after call on one stream (the word must be
    back at zero every time), two streams from two threads, and a hipGraph replayed three times."""
    import ctypes as C
    import threading
    import torch
    L, E = ec.lib(), ec._ffi
    rng = np.random.default_rng(11)
    big = (rng.random(9_000_011) < 0.37).astype(np.uint8)
    dbig = ec.Mask.new(big)
    try:
        for n, off in ((1, 0), (4096, 0), (65_536, 0), (65_537, 1), (131_073, 3), (1_000_003, 5), (9_000_000, 7)):
            want = (int(big[off:off + n].sum()), n - int(big[off:off + n].sum()))
            assert eco.mask_counts(big[off:off + n]) == want
            for mode in (1, 0, 2, 2):
                E.check(L.ec_tune_set(b"counts_one_launch", mode))
                assert dbig.shard(off, n).counts() == want, (n, off, mode)
        E.check(L.ec_tune_set(b"counts_one_launch", 2))
        # two host threads, each on its own stream (each stream has its own accumulator word)
        errs = []

        def worker(k):
            try:
                s = C.c_void_p()
                E.check(L.ec_stream_create(C.byref(s)))
                t, f = C.c_uint64(), C.c_uint64()
                for i in range(40):
                    n = 2_000_000 + 13 * i + k
                    E.check(L.ec_mask_counts(dbig.mem.ptr + k, n, C.byref(t), C.byref(f), s))
                    assert (t.value, f.value) == (int(big[k:k + n].sum()), n - int(big[k:k + n].sum()))
                E.check(L.ec_stream_destroy(s))
            except BaseException as e:  # noqa: BLE001
                errs.append(repr(e))
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        for t_ in ths: t_.start()
        for t_ in ths: t_.join()
        assert not errs, errs
        # captured: the kernel zeroes its word itself, so a replay finds it as the capture did
        cap = torch.cuda.Stream()
        E.check(L.ec_prepare_stream(cap.cuda_stream))
        out = torch.zeros(2, dtype=torch.int64, device="cuda")
        g = torch.cuda.CUDAGraph()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=cap):
            E.check(L.ec_mask_counts_device(dbig.mem.ptr, 5_000_000, out.data_ptr(), cap.cuda_stream))
        for _ in range(3):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert out.tolist() == [int(big[:5_000_000].sum()), 5_000_000 - int(big[:5_000_000].sum())]
    finally:
        E.check(L.ec_tune_set(b"counts_one_launch", 1))


def test_fill_under_every_occupancy_cap(ec):
    """`ec_fill` reserves unused LDS so that few workgroups per CU write (`write_lds_kb`, 64 KiB by default: two per CU); the knob changes
    the launch, never the cells — every cap, every cell width, ragged lengths, odd window offsets; and the knob reads back."""
    import ctypes as C
    L = ec.lib()
    v = C.c_int64(0)
    assert L.ec_stat_get(b"tune.write_lds_kb", C.byref(v)) == 0 and v.value == 64
    try:
        for kb in (0, 16, 48, 64, 1000):  # clamped to 64
            assert L.ec_tune_set(b"write_lds_kb", kb) == 0
            assert L.ec_stat_get(b"tune.write_lds_kb", C.byref(v)) == 0 and v.value == min(kb, 64)
            for ct, x in ((ec.UInt8, 201), (ec.Int16, -12345), (ec.Float32, 2.5), (ec.Float64, -0.0), (ec.UInt64, 2**63 + 5)):
                for n, off in ((100_003, 0), (4096, 3), (1, 0)):
                    big = ec.CellBuffer.fill(n + 8, ec.CellValue(ct, 0))
                    w = big.shard(off, n)
                    ec._ffi.check(L.ec_fill(ct, w.mem.ptr, n, C.byref(ec.CellValue(ct, x).to_ec()), ec.stream()))
                    got = big.to_numpy()
                    exp = np.zeros(n + 8, dtype=ec.NP_DTYPES[ct])
                    exp[off:off + n] = x
                    assert np.array_equal(bits_of(got), bits_of(exp)), (kb, ct, n, off)
    finally:
        L.ec_tune_set(b"write_lds_kb", 64)


def test_streams_threads_and_graph_capture(ec):
    """Re-entrancy: concurrent host threads on distinct streams (per-stream reduction scratch, per-thread
    device binding), and a chain of asynchronous calls captured into a hipGraph and replayed."""
    import ctypes as C
    import threading
    L = ec.lib()
    n = 1 << 20
    inputs = [(rand_cells(eco.U16, n, 600 + i), rand_cells(eco.U16, n, 700 + i)) for i in range(4)]
    results = [None] * 4

    def worker(i):
        s = C.c_void_p()
        ec._ffi.check(L.ec_stream_create(C.byref(s)))
        try:
            a, b = inputs[i]
            # pooled blocks are allocated on the stream that uses them (stream-ordered allocation)
            da, db = ec.DeviceMem(a.nbytes, stream=s), ec.DeviceMem(b.nbytes, stream=s)
            out, keys = ec.DeviceMem(n * 8, stream=s), ec.DeviceMem(16, stream=s)
            ec._ffi.check(L.ec_upload(da.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, s))
            ec._ffi.check(L.ec_upload(db.ptr, b.ctypes.data_as(C.c_void_p), b.nbytes, s))
            k = np.empty(2, np.int64)
            for _ in range(20):
                ec._ffi.check(L.ec_binop(ec.DIV, ec.UInt16, da.ptr, ec.UInt16, db.ptr, n, out.ptr, s))
                ec._ffi.check(L.ec_min_max_keys(ec.Float64, out.ptr, None, n, keys.ptr, s))
            ec._ffi.check(L.ec_download(k.ctypes.data_as(C.c_void_p), keys.ptr, 16, s))
            results[i] = (int(k[0]), int(k[1]))
            del da, db, out, keys  # back to the pool on `s`, before the stream goes away
        finally:
            ec._ffi.check(L.ec_stream_destroy(s))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    from erased_cells_hip import sharded
    for i, (a, b) in enumerate(inputs):
        mn, mx = sharded.combine_min_max_keys(ec.Float64, results[i])
        emn, emx = eco.f_min_max(eco.f_binop(eco.DIV, a, b))
        assert (mn.bits(), mx.bits()) == (emn.bits(), emx.bits()), i

    # hipGraph: capture NDVI (fused) + min_max keys on a torch stream, replay after changing the inputs
    import torch
    side = torch.cuda.Stream()
    ec._ffi.check(L.ec_prepare_stream(side.cuda_stream))
    nir, red = rand_cells(eco.U16, n, 801), rand_cells(eco.U16, n, 802)
    t_nir, t_red = torch.from_numpy(nir.view(np.int16)).cuda(), torch.from_numpy(red.view(np.int16)).cuda()
    t_out = torch.empty(n, dtype=torch.float64, device="cuda")
    t_keys = torch.zeros(2, dtype=torch.int64, device="cuda")
    dt4 = (C.c_uint8 * 4)(ec.UInt16, ec.UInt16, ec.UInt16, ec.UInt16)
    p4 = (C.c_void_p * 4)(t_nir.data_ptr(), t_red.data_ptr(), t_nir.data_ptr(), t_red.data_ptr())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        h = torch.cuda.current_stream().cuda_stream
        ec._ffi.check(L.ec_fused(ec.SUB, ec.DIV, ec.ADD, dt4, p4, None, n, t_out.data_ptr(), h))
        ec._ffi.check(L.ec_min_max_keys(ec.Float64, t_out.data_ptr(), None, n, t_keys.data_ptr(), h))
    for trial in range(2):
        if trial == 1:  # new inputs in the same buffers, same graph
            nir, red = rand_cells(eco.U16, n, 803), rand_cells(eco.U16, n, 804)
            t_nir.copy_(torch.from_numpy(nir.view(np.int16)))
            t_red.copy_(torch.from_numpy(red.view(np.int16)))
        g.replay()
        torch.cuda.synchronize()
        exp = eco.f_binop(eco.DIV, eco.f_binop(eco.SUB, nir, red), eco.f_binop(eco.ADD, nir, red))
        assert np.array_equal(bits_of(t_out.cpu().numpy()), bits_of(exp))
        k = t_keys.cpu().numpy()
        mn, mx = sharded.combine_min_max_keys(ec.Float64, (int(k[0]), int(k[1])))
        emn, emx = eco.f_min_max(exp)
        assert (mn.bits(), mx.bits()) == (emn.bits(), emx.bits())


def test_fused_chain_over_four_cell_types_in_a_hipgraph(ec):
    """Rounds 1-2 widened mixed-type operands into pooled temporaries first, so such a call could not be captured.  Every
    fused call is now one allocation-free launch: `(u8 - i16) / (f32 + f64)` with four masks is captured once and replayed
    on new contents of the same buffers."""
    import ctypes as C
    import torch
    L = ec.lib()
    n = (1 << 18) + 5
    types = (eco.U8, eco.I16, eco.F32, eco.F64)
    np_t = (np.uint8, np.int16, np.float32, np.float64)
    side = torch.cuda.Stream()
    bufs = [torch.empty(n * np.dtype(t).itemsize, dtype=torch.uint8, device="cuda") for t in np_t]
    masks = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
    out, om = torch.empty(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda")
    dt4 = (C.c_uint8 * 4)(*types)
    p4 = (C.c_void_p * 4)(*[b.data_ptr() for b in bufs])
    m4 = (C.c_void_p * 4)(*[m.data_ptr() for m in masks])

    def fill(seed):
        host = [rand_cells(t, n, seed + k) for k, t in enumerate(types)]
        hm = [rand_mask(n, seed + 10 + k) for k in range(4)]
        for b, h in zip(bufs, host):
            b.copy_(torch.from_numpy(h.view(np.uint8)))
        for m, h in zip(masks, hm):
            m.copy_(torch.from_numpy(h))
        return host, hm

    host, hm = fill(900)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        h = torch.cuda.current_stream().cuda_stream
        ec._ffi.check(L.ec_masked_fused(eco.SUB, eco.DIV, eco.ADD, dt4, p4, m4, None, n, out.data_ptr(), om.data_ptr(), h))
    for trial in range(2):
        if trial == 1:
            host, hm = fill(950)
        g.replay()
        torch.cuda.synchronize()
        e1, e2 = eco.f_binop(eco.SUB, host[0], host[1]), eco.f_binop(eco.ADD, host[2], host[3])
        exp = eco.f_binop(eco.DIV, e1, e2)
        from vectors import chain_loose
        assert_f64_bits_equal(out.cpu().numpy(), exp, nan_by_class_where=chain_loose(eco.SUB, host[0], host[1], eco.DIV, e1, e2, eco.ADD, host[2], host[3]))
        assert np.array_equal(om.cpu().numpy(), hm[0] & hm[1] & hm[2] & hm[3])


@pytest.mark.timeout(300)
def test_native_rccl_allreduce_of_reduction_payloads(ec):
    """ec_allreduce_min_max_keys / ec_allreduce_counts drive RCCL directly (no torch): a 1-rank
    communicator on this GPU must leave the payloads unchanged and decode to the oracle's answer."""
    import ctypes as C
    import os
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so.1")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        L = ec.lib()
        a = rand_cells(eco.F32, 200001, 501)
        m = rand_mask(200001, 502)
        d, dm = ec.CellBuffer.from_vec(a), ec.Mask.new(m)
        keys, counts = ec.DeviceMem(16), ec.DeviceMem(16)
        ec._ffi.check(L.ec_min_max_keys(ec.Float32, d.mem.ptr, dm.mem.ptr, d.len(), keys.ptr, None))
        ec._ffi.check(L.ec_allreduce_min_max_keys(comm, keys.ptr, None))
        ec._ffi.check(L.ec_mask_counts_device(dm.mem.ptr, dm.len(), counts.ptr, None))
        ec._ffi.check(L.ec_allreduce_counts(comm, counts.ptr, None))
        k, c = np.empty(2, np.int64), np.empty(2, np.uint64)
        ec._ffi.check(L.ec_download(k.ctypes.data_as(C.c_void_p), keys.ptr, 16, None))
        ec._ffi.check(L.ec_download(c.ctypes.data_as(C.c_void_p), counts.ptr, 16, None))
        from erased_cells_hip import sharded
        mn, mx = sharded.combine_min_max_keys(ec.Float32, (int(k[0]), int(k[1])))
        emn, emx = eco.f_min_max(a, m)
        assert (mn.bits(), mx.bits()) == (emn.bits(), emx.bits())
        assert (int(c[0]), int(c[1])) == eco.mask_counts(m)
        assert L.ec_allreduce_min_max_keys(None, keys.ptr, None) == ec._ffi.EC_ERR_ARG
    finally:
        rccl.ncclCommDestroy(comm)


def test_synthetic_generators_match_oracle(ec):
    """bench.py's device-side input generator == the oracle's."""
    import ctypes as C
    n = 100003
    a = ec.CellBuffer.empty(n, ec.UInt8)
    b = ec.CellBuffer.empty(n, ec.UInt16)
    ec._ffi.check(ec.lib().ec_synth_fill(ec.UInt8, a.mem.ptr, n, 0x5EED0001, 5, 0.0, 255.0, None))
    ec._ffi.check(ec.lib().ec_synth_fill(ec.UInt16, b.mem.ptr, n, 0x5EED0002, 5, 1.0, 65535.0, None))
    assert np.array_equal(a.to_numpy(), eco.fill_u8(n, 0x5EED0001, base=5))
    assert np.array_equal(b.to_numpy(), eco.fill_u16(n, 0x5EED0002, base=5, lo=1))
    m = ec.Mask.empty(n)
    ec._ffi.check(ec.lib().ec_synth_mask(m.mem.ptr, n, 0x5EED0013, 0, 30, None))
    exp = np.array([eco.splitmix64(0x5EED0013 ^ i) % 100 >= 30 for i in range(2000)], np.uint8)
    assert np.array_equal(m.to_numpy()[:2000], exp)
    t, f = m.counts()
    assert abs(f / n - 0.30) < 0.01
