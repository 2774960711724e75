"""Parity at BASELINE.json's full sizes (configs 2-4) on one MI355X.

Config 2 and 3 are compared cell for cell against the oracle's typed loops
(the GPU box has the host memory for it); config 4 (65536² u16, 8.6 GB) uses
size-independent properties: planted extremes, shard-combine == whole, and a
checksum-of-counts identity for masks.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import eco
from vectors import bits_of

pytestmark = pytest.mark.gpu

SIDE = 16384
N = SIDE * SIDE


@pytest.fixture(scope="module")
def ec():
    import erased_cells_hip as ec
    ec.init(0)
    return ec


def _chk(ec, st):
    ec._ffi.check(st)


def _assert_same_bits(got, exp, chunk=1 << 26):
    assert got.shape == exp.shape and got.dtype == exp.dtype
    g, e = bits_of(got), bits_of(exp)
    for i in range(0, g.size, chunk):
        if not np.array_equal(g[i:i + chunk], e[i:i + chunk]):
            j = i + int(np.flatnonzero(g[i:i + chunk] != e[i:i + chunk])[0])
            raise AssertionError(f"first difference at cell {j}: got {got[j]!r} expected {exp[j]!r}")


def test_config2_divide_u8_u16_16384sq_bit_exact(ec):
    """16384 x 16384 u8 ÷ u16 -> f64, bench.py's exact inputs, both kernel variants, every cell."""
    L = ec.lib()
    a, b = ec.CellBuffer.empty(N, ec.UInt8), ec.CellBuffer.empty(N, ec.UInt16)
    _chk(ec, L.ec_synth_fill(ec.UInt8, a.mem.ptr, N, 0x5EED0001, 0, 0.0, 255.0, None))
    _chk(ec, L.ec_synth_fill(ec.UInt16, b.mem.ptr, N, 0x5EED0002, 0, 1.0, 65535.0, None))
    eco.set_threads(16)
    try:
        ha, hb = eco.fill_u8(N, 0x5EED0001), eco.fill_u16(N, 0x5EED0002, lo=1)
        assert np.array_equal(a.to_numpy(), ha) and np.array_equal(b.to_numpy(), hb)
        exp = eco.f_binop(eco.DIV, ha, hb)
    finally:
        eco.set_threads(0)
    # the reference-shaped oracle on a slice pins the typed loops at this size too
    assert np.array_equal(bits_of(eco.binop(eco.DIV, ha[-5000:], hb[-5000:])), bits_of(exp[-5000:]))
    for variant in (0, 1):
        L.ec_tune_set(b"binop_variant", variant)
        try:
            out = a / b
            assert out.cell_type() == ec.Float64 and out.len() == N
            _assert_same_bits(out.to_numpy(), exp)
            mn, mx = out.min_max()
            assert (mn.value, mx.value) == (exp.min(), exp.max())
        finally:
            L.ec_tune_set(b"binop_variant", -1)  # the default: by rule
    # the 8 row-block shards of §8e, computed separately, tile the same result
    from erased_cells_hip import sharded
    for g in (0, 3, 7):
        off, ln = sharded.shard_range(SIDE, SIDE, g, 8)
        part = a.shard(off, ln) / b.shard(off, ln)
        _assert_same_bits(part.to_numpy(), exp[off:off + ln])


def test_config3_masked_f32_chain_16384sq(ec):
    """(a + b) * c on MaskedCellBuffer f32 with 30 % nodata masks, eager, every cell and mask byte."""
    L = ec.lib()
    bufs, masks = [], []
    for i in range(3):
        bf, mk = ec.CellBuffer.empty(N, ec.Float32), ec.Mask.empty(N)
        _chk(ec, L.ec_synth_fill(ec.Float32, bf.mem.ptr, N, 0x5EED0003 + i, 0, -1000.0, 1000.0, None))
        _chk(ec, L.ec_synth_mask(mk.mem.ptr, N, 0x5EED0013 + i, 0, 30, None))
        bufs.append(bf)
        masks.append(mk)
    ma, mb, mc = (ec.MaskedCellBuffer(b, m) for b, m in zip(bufs, masks))
    r = (ma + mb) * mc
    ha, hb, hc = (b.to_numpy() for b in bufs)
    hm = [m.to_numpy() for m in masks]
    assert ha.min() >= -1000.0 and ha.max() <= 1000.0 and abs(float(ha.mean())) < 1.0  # f32 rounding can reach the open end
    eco.set_threads(16)
    try:
        t = eco.f_binop(eco.ADD, ha, hb)
        exp = eco.f_binop(eco.MUL, t, hc)
    finally:
        eco.set_threads(0)
    _assert_same_bits(r.buffer().to_numpy(), exp)
    em = hm[0] & hm[1] & hm[2]
    assert np.array_equal(r.mask().to_numpy(), em)
    assert r.counts() == (int(em.sum()), int(N - em.sum()))
    assert abs(r.counts()[0] / N - 0.7 ** 3) < 1e-3
    mn, mx = r.min_max()
    valid = exp[em.astype(bool)]
    assert (mn.value, mx.value) == (valid.min(), valid.max())
    # mask built from a sentinel: inject the canonical NaN where mask a is false, rebuild it on device
    inj = ha.copy()
    inj[hm[0] == 0] = np.float32(np.nan)
    rebuilt = ec.MaskedCellBuffer.from_vec_with_nodata(inj, ec.NoData.default())
    assert np.array_equal(rebuilt.mask().to_numpy(), hm[0])
    # scalar variant of examples/masked.rs:12: (a + b) * 2.0 keeps the AND-ed mask
    r2 = (ma + mb) * 2.0
    _assert_same_bits(r2.buffer().to_numpy(), eco.f_binop_scalar(eco.MUL, t, eco.Value.of(eco.F64, 2.0)))
    assert np.array_equal(r2.mask().to_numpy(), hm[0] & hm[1])


def test_config4_min_max_u16_65536sq_planted_and_sharded(ec):
    """65536² u16 (8.6 GB): planted global extremes in different row-blocks; the 8 shard key pairs,
    MAX-combined exactly as the RCCL all-reduce does, equal the single-pass result."""
    from erased_cells_hip import sharded
    L = ec.lib()
    side = 65536
    n = side * side
    buf = ec.CellBuffer.empty(n, ec.UInt16)
    _chk(ec, L.ec_synth_fill(ec.UInt16, buf.mem.ptr, n, 0x5EED0006, 0, 1.0, 65534.0, None))
    mn, mx = buf.min_max()
    assert (mn.ct, mn.value, mx.value) == (ec.UInt16, 1, 65534)
    shards = [sharded.shard_range(side, side, g, 8) for g in range(8)]
    assert all(ln == 8192 * side for _, ln in shards)
    lo_at = shards[5][0] + 123_456_789 % shards[5][1]
    hi_at = shards[2][0] + 987_654_321 % shards[2][1]
    buf.put(lo_at, ec.CellValue(ec.UInt16, 0))
    buf.put(hi_at, ec.CellValue(ec.UInt16, 65535))
    mn, mx = buf.min_max()
    assert (mn.value, mx.value) == (0, 65535)
    keys_dev = ec.DeviceMem(16)
    k = np.empty(2, np.int64)
    comb = None
    for g, (off, ln) in enumerate(shards):
        part = buf.shard(off, ln)
        _chk(ec, L.ec_min_max_keys(ec.UInt16, part.mem.ptr, None, ln, keys_dev.ptr, None))
        _chk(ec, L.ec_download(k.ctypes.data_as(C.c_void_p), keys_dev.ptr, 16, None))
        lmn, lmx = sharded.combine_min_max_keys(ec.UInt16, (int(k[0]), int(k[1])))
        assert lmn.value == (0 if g == 5 else 1) and lmx.value == (65535 if g == 2 else 65534)
        comb = (int(k[0]), int(k[1])) if comb is None else (max(comb[0], int(k[0])), max(comb[1], int(k[1])))
    gmn, gmx = sharded.combine_min_max_keys(ec.UInt16, comb)
    assert (gmn.value, gmx.value) == (0, 65535)
    # masked: hide both plants -> back to (1, 65534); counts identity on a 4.3 G-cell mask
    mask = ec.Mask.empty(n)
    _chk(ec, L.ec_synth_mask(mask.mem.ptr, n, 0x5EED0016, 0, 30, None))
    mask.put(lo_at, False)
    mask.put(hi_at, False)
    mmn, mmx = ec.MaskedCellBuffer(buf, mask).min_max()
    assert (mmn.value, mmx.value) == (1, 65534)
    t, f = mask.counts()
    assert t + f == n and abs(f / n - 0.30) < 1e-3
    tot = sum(mask.shard(off, ln).counts()[0] for off, ln in shards)
    assert tot == t
    nt, nf = (~mask).counts()
    assert (nt, nf) == (f, t)


def test_indexing_beyond_2_32_cells(ec):
    """More than 2^32 cells in one buffer (a 65537 x 65537 u8 raster, odd length): 64-bit indexing in the
    element-wise, map and reduction kernels; spot-checked on slices against the oracle."""
    L = ec.lib()
    n = 65537 * 65537  # 4,295,098,369 cells
    a, b = ec.CellBuffer.empty(n, ec.UInt8), ec.CellBuffer.empty(n, ec.UInt8)
    _chk(ec, L.ec_synth_fill(ec.UInt8, a.mem.ptr, n, 0x5EED0021, 0, 0.0, 255.0, None))
    _chk(ec, L.ec_synth_fill(ec.UInt8, b.mem.ptr, n, 0x5EED0022, 0, 1.0, 255.0, None))
    out = a / b                      # 34 GB of f64
    wide = a.convert(ec.UInt16)      # k_map
    neg = -a                         # u8 -> i16
    spots = [0, (1 << 32) - 70000, (1 << 32) - 5, n - 100003]
    for off in spots:
        ln = min(100003, n - off)
        ha = eco.fill_u8(ln, 0x5EED0021, base=off)
        hb = eco.fill_u8(ln, 0x5EED0022, base=off, lo=1)
        assert np.array_equal(a.shard(off, ln).to_numpy(), ha)
        _assert_same_bits(out.shard(off, ln).to_numpy(), eco.f_binop(eco.DIV, ha, hb))
        assert np.array_equal(wide.shard(off, ln).to_numpy(), ha.astype(np.uint16))
        assert np.array_equal(neg.shard(off, ln).to_numpy(), -ha.astype(np.int16))
    mn, mx = out.min_max()
    assert (mn.value, mx.value) == (0.0, 255.0)
    a.put(n - 1, ec.CellValue(ec.UInt8, 0))
    b.put(n - 1, ec.CellValue(ec.UInt8, 1))
    m = ec.mask_from_nodata(a, ec.NoData.default())     # zeros are nodata
    t, f = m.counts()
    assert t + f == n and abs(f / n - 1 / 256) < 1e-4
    assert ec.MaskedCellBuffer(a, m).min_max()[0].value == 1


def test_evi_expression_program_16384sq_three_ways(ec):
    """An eight-operator tree at raster scale (three 16384² u16 bands, bench.py's `--workload evi` inputs): the built-in kernel
    of the ahead-of-time catalogue (what runs by default), the interpreter kernel, the program compiled for itself and the eager
    chain of eight operators agree on every cell (compared on the device), and the ends and the middle of the raster agree
    with the oracle's step-by-step evaluation."""
    L, E, P = ec.lib(), ec._ffi, ec.fused
    bands, seeds = [], (0x5EED0031, 0x5EED0032, 0x5EED0033)
    ranges = [(2000 + 3000 * (2 - i), 20000 + 10000 * (2 - i)) for i in range(3)]
    for seed, (lo, hi) in zip(seeds, ranges):
        b = ec.CellBuffer.empty(N, ec.UInt16)
        _chk(ec, L.ec_synth_fill(ec.UInt16, b.mem.ptr, N, seed, 0, float(lo), float(hi), None))
        bands.append(b)
    nir, red, blue = bands
    S, R, K = (lambda k: k), (lambda k: 4 + k), (lambda k: 8 + k)
    prog = [(ec.SUB, S(0), S(1), 0), (ec.MUL, R(0), K(0), 0), (ec.MUL, S(1), K(1), 1), (ec.ADD, S(0), R(1), 1),
            (ec.MUL, S(2), K(2), 2), (ec.SUB, R(1), R(2), 1), (ec.ADD, R(1), K(3), 1), (ec.DIV, R(0), R(1), 0)]
    ks = [2.5, 6.0, 7.5, 1.0]
    def stat(key):
        v = C.c_int64(0)
        _chk(ec, L.ec_stat_get(key, C.byref(v)))
        return v.value

    try:
        L.ec_tune_set(b"expr_jit", 0)
        f0 = stat(b"expr_fixed_launches")
        builtin = P.program(bands, ks, prog)
        assert stat(b"expr_fixed_launches") == f0 + 1
        L.ec_tune_set(b"expr_fixed", 0)
        i0 = stat(b"expr_interp_launches")
        interpreted = P.program(bands, ks, prog)
        assert stat(b"expr_interp_launches") == i0 + 1
        L.ec_tune_set(b"expr_jit", 2)
        compiled = P.program(bands, ks, prog)
    finally:
        L.ec_tune_set(b"expr_jit", 1)
        L.ec_tune_set(b"expr_fixed", 1)
    assert interpreted == compiled                      # ec_buffer_cmp: first differing cell on the device, none
    assert builtin == compiled
    del builtin
    eager = ((nir - red) * 2.5) / (((nir + red * 6.0) - blue * 7.5) + 1.0)
    assert eager == compiled
    del interpreted, eager
    got = compiled.to_numpy()
    for lo in (0, N // 2 - 12345, N - (1 << 20)):
        hs = [eco.fill_u16(1 << 20, seed, base=lo, lo=r[0], hi=r[1]) for seed, r in zip(seeds, ranges)]
        assert np.array_equal(hs[0], nir.shard(lo, 1 << 20).to_numpy())
        f = lambda op, a, b: eco.f_binop(op, a, b if isinstance(b, np.ndarray) else np.full(1 << 20, float(b)))  # noqa: E731
        exp = f(eco.DIV, f(eco.MUL, f(eco.SUB, hs[0], hs[1]), 2.5),
                f(eco.ADD, f(eco.SUB, f(eco.ADD, hs[0], f(eco.MUL, hs[1], 6.0)), f(eco.MUL, hs[2], 7.5)), 1.0))
        _assert_same_bits(got[lo:lo + (1 << 20)], exp)
