"""The oracle's two forms (reference-shaped eco_* and typed-loop ecof_*) and an
independent numpy evaluation must agree bit for bit on seeded inputs that
include every type's extremes, values where `as f64` rounds, signed zeros,
infinities and NaNs.  Also pins the claim binop ≡ (l as f64) op (r as f64)
 on all 100 type pairs × 4 ops.  No GPU."""
import numpy as np
import pytest

from oracle import eco
from oracle.eco import ADD, DIV, F32, F64, MUL, ND_DEFAULT, ND_VALUE, NTYPES, SUB, Value
from vectors import assert_f64_bits_equal, bits_of, rand_cells, rand_mask

N = 777  # not a multiple of any vector width
NPOP = {ADD: np.add, SUB: np.subtract, MUL: np.multiply, DIV: np.divide}


@pytest.mark.parametrize("lct", range(NTYPES))
def test_binop_forms_agree_all_pairs(lct):
    for rct in range(NTYPES):
        l, r = rand_cells(lct, N, 1), rand_cells(rct, N, 2)
        for op in (ADD, SUB, MUL, DIV):
            ref = eco.binop(op, l, r)
            fast = eco.f_binop(op, l, r)
            assert ref.dtype == np.float64
            with np.errstate(all="ignore"):  # signalling-NaN cells raise "invalid" in the f32 -> f64 cast
                both_nan = np.isnan(l.astype(np.float64)) & np.isnan(r.astype(np.float64))
            assert_f64_bits_equal(fast, ref, nan_by_class_where=both_nan)
            with np.errstate(all="ignore"):
                npv = NPOP[op](l.astype(np.float64), r.astype(np.float64))
            # numpy: same IEEE op; NaN sign/payload of its SIMD loops is not pinned -> class only
            assert_f64_bits_equal(npv, ref, nan_by_class_where=np.ones(N, bool))


@pytest.mark.parametrize("lct", range(NTYPES))
def test_binop_scalar_forms_agree(lct):
    l = rand_cells(lct, N, 3)
    for sct in range(NTYPES):
        for sval in (rand_cells(sct, 4, 9, specials=False)[0], 2, 0):
            s = Value.of(sct, sval)
            for op in (ADD, SUB, MUL, DIV):
                with np.errstate(all="ignore"):
                    lnan = np.isnan(l.astype(np.float64))
                assert_f64_bits_equal(eco.f_binop_scalar(op, l, s), eco.binop_scalar(op, l, s), nan_by_class_where=lnan)


@pytest.mark.parametrize("ct", range(NTYPES))
def test_neg_convert_minmax_mask_forms_agree(ct):
    a = rand_cells(ct, N, 4)
    assert np.array_equal(bits_of(eco.f_neg(a)), bits_of(eco.neg(a)))
    assert eco.f_neg(a).dtype == eco.neg(a).dtype
    for dt in range(NTYPES):
        if eco.can_fit_into(ct, dt):
            assert np.array_equal(bits_of(eco.f_convert(a, dt)), bits_of(eco.convert(a, dt)))
    for mask in (None, rand_mask(N, 5), np.zeros(N, np.uint8)):
        (m1, x1), (m2, x2) = eco.min_max(a, mask), eco.f_min_max(a, mask)
        assert (m1.ct, m1.bits(), x1.ct, x1.bits()) == (m2.ct, m2.bits(), x2.ct, x2.bits())
    for nd in (eco.nodata_value(ND_DEFAULT, ct), Value.of(ct, a[7]), None):
        kind = ND_VALUE if nd is not None else 0
        ref = eco.mask_from_nodata(a, kind, nd)
        assert np.array_equal(eco.f_mask_from_nodata(a, nd), ref)
        m = rand_mask(N, 6)
        assert np.array_equal(bits_of(eco.f_mask_select(a, m, nd)), bits_of(eco.mask_select(a, m, kind, nd)))


def test_total_order_min_max_on_floats():
    for ct, dt in ((F32, np.float32), (F64, np.float64)):
        u = np.uint32 if ct == F32 else np.uint64
        neg_nan = np.array([0xFFC00000 if ct == F32 else 0xFFF8000000000000], dtype=u).view(dt)[0]
        a = np.array([1.0, np.inf, -np.inf, np.nan, neg_nan, -0.0, 0.0], dtype=dt)
        mn, mx = eco.min_max(a)
        assert mn.bits() == bits_of(np.array([neg_nan]))[0]  # -NaN sorts below -inf
        assert np.isnan(mx.get()) and not np.signbit(mx.get())  # +NaN above +inf
        # finite sentinels (ctype.rs:158-179): min over [+inf] is f::MAX
        mn, mx = eco.min_max(np.array([np.inf], dtype=dt))
        assert mn.get() == np.finfo(dt).max and mx.get() == np.inf
        mn, mx = eco.min_max(np.array([-np.inf], dtype=dt))
        assert mn.get() == -np.inf and mx.get() == np.finfo(dt).min
        mn, mx = eco.min_max(np.array([], dtype=dt))
        assert mn.get() == np.finfo(dt).max and mx.get() == np.finfo(dt).min
        mn, mx = eco.min_max(np.array([-0.0, 0.0], dtype=dt))
        assert np.signbit(mn.get()) and not np.signbit(mx.get())


def test_i64_u64_to_f64_rounding_is_rne():
    # parity by language spec (Rust `as`), unpinned by reference tests
    u = np.array([2**53, 2**53 + 1, 2**53 + 2, 2**53 + 3, 2**64 - 1, 2**63 + 1025], dtype=np.uint64)
    exp = [float(2**53), float(2**53), float(2**53 + 2), float(2**53 + 4), float(2**64), float(2**63 + 2048)]
    assert eco.convert(u, F64).tolist() == exp
    i = np.array([-(2**53) - 1, -(2**63), 2**63 - 1], dtype=np.int64)
    assert eco.convert(i, F64).tolist() == [-float(2**53), -float(2**63), float(2**63)]
    assert eco.binop(ADD, u, np.zeros(6, np.uint8)).tolist() == exp


def test_x86_default_nan_is_negative():
    # documents the host-FPU dependence recorded in ec_oracle.h / DESIGN.md "NaN policy"
    z = np.zeros(1, np.uint8)
    r = eco.binop(DIV, z, z)
    assert bits_of(r)[0] == 0xFFF8000000000000
    r = eco.binop(SUB, np.array([np.inf]), np.array([np.inf]))
    assert bits_of(r)[0] == 0xFFF8000000000000
    # propagated NaNs keep sign and payload (quieted)
    q = np.array([0x7FF0000000000001], dtype=np.uint64).view(np.float64)
    r = eco.binop(ADD, q, np.ones(1))
    assert bits_of(r)[0] == 0x7FF8000000000001
