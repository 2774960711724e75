"""Every kernel instantiation the library ships is executed at least once against the oracle.

The parity tests walk the default path over all type pairs; this file sweeps the remaining template
instantiations — the LDS-staged variant, the masked kernels, all 624 fused kernels (one per quadruple of operand
load classes) under every op triple and cell kind, the map kernels at every tile depth and the cell-wise comparison path — on small inputs.  (Which kernels a run
executed can be listed with `rocprofv3 --kernel-trace --stats -- python3 -m pytest tests -m gpu`.)
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import eco
from vectors import assert_f64_bits_equal, bits_of, chain_loose, rand_cells, rand_mask

pytestmark = pytest.mark.gpu

OPS = [eco.ADD, eco.SUB, eco.MUL, eco.DIV]
NT = eco.NTYPES
N = 2600  # a few wave tiles + a ragged tail


@pytest.fixture(scope="module")
def ec():
    import erased_cells_hip as ec
    ec.init(0)
    return ec


@pytest.fixture(scope="module")
def pool(ec):
    host = {ct: rand_cells(ct, N + 8, 4100 + ct) for ct in range(NT)}
    dev = {ct: ec.CellBuffer.from_vec(a) for ct, a in host.items()}
    m = [rand_mask(N + 8, 4200 + k) for k in range(2)]
    return host, dev, m, [ec.Mask.new(x) for x in m]


def _both_nan(l, r):
    with np.errstate(all="ignore"):
        return np.isnan(l.astype(np.float64)) & np.isnan(r.astype(np.float64))


def _loose(op, l, r):
    return _both_nan(l, r) if op in (eco.ADD, eco.MUL) else None


@pytest.mark.parametrize("variant,vector,off", [(1, 1, 0), (0, 0, 1)], ids=["lds-staged", "cellwise"])
def test_binop_and_masked_binop_every_pair(ec, pool, variant, vector, off):
    """k_binop_lds / k_masked_binop<LDS> for every staged pair, and the cell-wise kernels for every pair
    (windows one cell into the buffers with the vector path switched off)."""
    host, dev, m, dm = pool
    L = ec.lib()
    L.ec_tune_set(b"binop_variant", variant)
    L.ec_tune_set(b"unaligned_vector", vector)
    try:
        for lt in range(NT):
            for rt in range(NT):
                l, r = host[lt][off:off + N], host[rt][off:off + N]
                dl, dr = dev[lt].shard(off, N), dev[rt].shard(off, N)
                ml = ec.MaskedCellBuffer(dl, dm[0].shard(off, N))
                mr = ec.MaskedCellBuffer(dr, dm[1].shard(off, N))
                for op in OPS:
                    exp = eco.f_binop(op, l, r)
                    assert_f64_bits_equal(dl._binop(op, dr).to_numpy(), exp, nan_by_class_where=_loose(op, l, r))
                    got = ml._binop(op, mr)
                    assert_f64_bits_equal(got.buffer().to_numpy(), exp, nan_by_class_where=_loose(op, l, r))
                    assert np.array_equal(got.mask().to_numpy(), m[0][off:off + N] & m[1][off:off + N])
            if vector == 0:  # scalar cell-wise kernels
                for op in OPS:
                    got = dev[lt].shard(off, N)._binop(op, 3)
                    assert_f64_bits_equal(got.to_numpy(), eco.f_binop_scalar(op, host[lt][off:off + N], eco.Value.of(eco.I32, 3)))
    finally:
        L.ec_tune_set(b"binop_variant", -1)  # the default: by rule
        L.ec_tune_set(b"unaligned_vector", 1)


def test_masked_binop_direct_every_pair(ec, pool):
    host, dev, m, dm = pool
    for lt in range(NT):
        ml = ec.MaskedCellBuffer(dev[lt].shard(0, N), dm[0].shard(0, N))
        for rt in range(NT):
            mr = ec.MaskedCellBuffer(dev[rt].shard(0, N), dm[1].shard(0, N))
            l, r = host[lt][:N], host[rt][:N]
            for op in OPS:
                got = ml._binop(op, mr)
                assert_f64_bits_equal(got.buffer().to_numpy(), eco.f_binop(op, l, r), nan_by_class_where=_loose(op, l, r))
                assert np.array_equal(got.mask().to_numpy(), m[0][:N] & m[1][:N])


def _oracle_chain(o1, hx, hy, o2, hz, o3=None, hw=None):
    """The eager chain on the oracle's typed loops + the cells whose NaN bits the reference leaves open."""
    e1 = eco.f_binop(o1, hx, hy)
    e2 = eco.f_binop(o3, hz, hw) if o3 is not None else hz
    eo = eco.f_binop(o2, e1, e2)
    return eo, chain_loose(o1, hx, hy, o2, e1, e2, o3, hz if o3 is not None else None, hw)


@pytest.mark.parametrize("ct", range(NT))
def test_fused_every_op_triple(ec, pool, ct):
    """All 80 (o1, o2, o3) op triples on operands of one cell type (k_fused_any<c,c,0,0> with the aliases of this
    call; the op arms are launch-uniform switches): bit-identical to the eager HIP chain,
    and to the ORACLE's chain under the same rule as test_binop_all_pairs_bit_exact (NaNs bit for bit, except cells
    where both operands of a commutative step are NaN)."""
    host, dev, _, _ = pool
    x, y = dev[ct].shard(0, N), dev[ct].shard(4, N)
    hx, hy = host[ct][:N], host[ct][4:4 + N]
    for o1 in OPS:
        t1 = x._binop(o1, y)
        for o2 in OPS:
            for o3 in OPS + [ec.fused.OP_NONE]:
                if o3 == ec.fused.OP_NONE:
                    got = ec.fused.expr(x, o1, y, o2, y)
                    exp = t1._binop(o2, y)
                    eo, loose = _oracle_chain(o1, hx, hy, o2, hy)
                else:
                    got = ec.fused.expr(x, o1, y, o2, y, o3, x)
                    exp = t1._binop(o2, y._binop(o3, x))
                    eo, loose = _oracle_chain(o1, hx, hy, o2, hy, o3, hx)
                g = got.to_numpy()
                assert np.array_equal(bits_of(g), bits_of(exp.to_numpy())), (ct, o1, o2, o3)
                assert_f64_bits_equal(g, eo, nan_by_class_where=loose)


# every ordered pair of distinct cell types (rounds 1-2 had one-pass kernels for 12 of these 90)
MIXED_PAIRS = [(a, b) for a in range(NT) for b in range(NT) if a != b]


def _pool_allocs(ec):
    v = C.c_int64()
    ec._ffi.check(ec.lib().ec_stat_get(b"pool_allocs", C.byref(v)))
    return v.value


@pytest.mark.parametrize("pair", MIXED_PAIRS, ids=lambda p: f"{eco.CT_NAMES[p[0]]}-{eco.CT_NAMES[p[1]]}")
def test_fused_mixed_every_instantiation(ec, pool, pair):
    """Every ordered pair of distinct cell types through the one-pass kernel (k_fused_any, ec_fused_any.hpp): A B A B with
    every op triple; A A B, A B A, A B B with every op pair: against the oracle's chain (strict NaN rule), against the
    eager HIP chain, against the convert-then-fuse path — and the one-pass path must not allocate."""
    host, dev, _, _ = pool
    A, B = pair
    L = ec.lib()
    a0, a1, b0, b1 = dev[A].shard(0, N), dev[A].shard(3, N), dev[B].shard(1, N), dev[B].shard(5, N)
    ha0, ha1, hb0, hb1 = host[A][:N], host[A][3:3 + N], host[B][1:1 + N], host[B][5:5 + N]
    out = ec.CellBuffer.empty(N, ec.Float64)  # result buffer made up front: only the kernel's own allocations would count

    def run(ops, o1, o2, o3):
        dt = (C.c_uint8 * 4)(*[o.ct for o in ops], *([0] * (4 - len(ops))))
        p = (C.c_void_p * 4)(*[o.mem.ptr for o in ops], *([None] * (4 - len(ops))))
        ec._ffi.check(L.ec_fused(o1, o2, o3, dt, p, None, N, out.mem.ptr, ec.stream()))
        return out.to_numpy()

    cases = [((a0, b0, a0, b0), (ha0, hb0, ha0, hb0)),   # A B A B, NDVI-shaped aliases
             ((a0, b0, a1, b1), (ha0, hb0, ha1, hb1))]   # A B A B, four distinct streams
    for o1 in OPS:
        for o2 in OPS:
            for o3 in OPS:
                for ops, h in cases[: 1 if (o1 + o2 + o3) % 2 else 2]:
                    before = _pool_allocs(ec)
                    g = run(ops, o1, o2, o3)
                    assert _pool_allocs(ec) == before, "the one-pass mixed kernel must not allocate"
                    eo, loose = _oracle_chain(o1, h[0], h[1], o2, h[2], o3, h[3])
                    assert_f64_bits_equal(g, eo, nan_by_class_where=loose)
            for ops, h in (((a0, a1, b0), (ha0, ha1, hb0)), ((a0, b0, a1), (ha0, hb0, ha1)), ((a0, b0, b1), (ha0, hb0, hb1)),
                           ((a0, b0, b0), (ha0, hb0, hb0))):
                before = _pool_allocs(ec)
                g = run(ops, o1, o2, ec.fused.OP_NONE)
                assert _pool_allocs(ec) == before
                eo, loose = _oracle_chain(o1, h[0], h[1], o2, h[2])
                assert_f64_bits_equal(g, eo, nan_by_class_where=loose)
    # the same calls through convert-then-fuse and through the eager chain give the same bits
    try:
        for (o1, o2, o3) in ((eco.SUB, eco.DIV, eco.ADD), (eco.MUL, eco.SUB, eco.DIV), (eco.ADD, eco.MUL, ec.fused.OP_NONE)):
            ops = (a0, b0, a1, b1) if o3 != ec.fused.OP_NONE else (a0, a1, b0)
            L.ec_tune_set(b"fused_mixed", 1)
            g1 = run(ops, o1, o2, o3).copy()
            L.ec_tune_set(b"fused_mixed", 0)
            before = _pool_allocs(ec)
            g0 = run(ops, o1, o2, o3).copy()
            assert _pool_allocs(ec) > before  # the fallback does draw temporaries
            t1 = ops[0]._binop(o1, ops[1])
            eager = t1._binop(o2, ops[2]._binop(o3, ops[3]) if o3 != ec.fused.OP_NONE else ops[2]).to_numpy()
            assert np.array_equal(bits_of(g1), bits_of(g0)) and np.array_equal(bits_of(g1), bits_of(eager))
    finally:
        L.ec_tune_set(b"fused_mixed", 1)
    # scalars fit any slot; masks ride along; windows at odd offsets and ragged lengths
    for n, off in ((1, 0), (2, 1), (513, 1), (2049, 3)):
        x, y = dev[A].shard(off, n), dev[B].shard(off + 1, n)
        hx, hy = host[A][off:off + n], host[B][off + 1:off + 1 + n]
        got = ec.fused.expr(x, eco.SUB, y, eco.DIV, x, eco.ADD, 2.5).to_numpy()   # A B A s
        assert np.array_equal(bits_of(got), bits_of(((x - y) / (x + 2.5)).to_numpy()))
        got = ec.fused.expr(x, eco.MUL, 3, eco.ADD, y).to_numpy()                  # A s B
        assert np.array_equal(bits_of(got), bits_of(((x * 3) + y).to_numpy()))
        got = ec.fused.ndvi(x, y).to_numpy()
        eo, loose = _oracle_chain(eco.SUB, hx, hy, eco.DIV, hx, eco.ADD, hy)
        assert_f64_bits_equal(got, eo, nan_by_class_where=loose)


@pytest.mark.parametrize("map_u,vector,off", [(1, 1, 0), (2, 1, 0), (4, 1, 0), (2, 0, 1)],
                         ids=["tile-depth-1", "tile-depth-2", "tile-depth-4", "cellwise"])
def test_map_kernels_every_instantiation(ec, pool, map_u, vector, off):
    """convert (every legal pair), neg, fill, mask_from_nodata, mask_select, mask logic and the reductions at every
    tile depth of k_map, and through the cell-wise kernels."""
    host, dev, m, dm = pool
    L = ec.lib()
    L.ec_tune_set(b"map_u", map_u)
    L.ec_tune_set(b"unaligned_vector", vector)
    try:
        ma, mb = dm[0].shard(off, N), dm[1].shard(off, N)
        ha, hb = m[0][off:off + N], m[1][off:off + N]
        assert np.array_equal((ma & mb).to_numpy(), eco.mask_and(ha, hb))
        assert np.array_equal((ma | mb).to_numpy(), eco.mask_or(ha, hb))
        assert np.array_equal((~ma).to_numpy(), eco.mask_not(ha))
        assert ma.counts() == eco.mask_counts(ha)
        for ct in range(NT):
            a, d = host[ct][off:off + N], dev[ct].shard(off, N)
            for dst in range(NT):
                if dst != ct and eco.can_fit_into(ct, dst):
                    assert np.array_equal(bits_of(d.convert(dst).to_numpy()), bits_of(eco.f_convert(a, dst))), (ct, dst)
            assert np.array_equal(bits_of((-d).to_numpy()), bits_of(eco.f_neg(a)))
            nd = eco.nodata_value(eco.ND_DEFAULT, ct)
            assert np.array_equal(ec.mask_from_nodata(d, ec.NoData.default()).to_numpy(), eco.f_mask_from_nodata(a, nd))
            sel = ec.MaskedCellBuffer(d, mb).to_vec_with_nodata(ct, ec.NoData.default())
            assert np.array_equal(bits_of(sel), bits_of(eco.f_mask_select(a, hb, nd)))
            for mask, dmask in ((None, None), (ha, ma)):
                got = d.min_max() if mask is None else ec.MaskedCellBuffer(d, dmask).min_max()
                exp = eco.f_min_max(a, mask)
                assert (got[0].bits(), got[1].bits()) == (exp[0].bits(), exp[1].bits())
            b = a.copy()
            b[N - 3] = a[1]
            assert d.cmp(ec.CellBuffer.from_vec(b)) == eco.buffer_cmp(a, b)
            f = ec.CellBuffer.fill(N, ec.CellValue(ct, a[5]))
            assert np.array_equal(bits_of(f.to_numpy()), bits_of(np.full(N, a[5], dtype=a.dtype)))
            w = ec.CellBuffer.empty(N + 8, ct).shard(off, N)  # fill into a window (the cell-wise fill when off = 1)
            v = ec.CellValue(ct, a[6]).to_ec()
            ec._ffi.check(L.ec_fill(ct, w.mem.ptr, N, C.byref(v), None))
            assert np.array_equal(bits_of(w.to_numpy()), bits_of(np.full(N, a[6], dtype=a.dtype)))
        # the device-side input generators of bench.py / the tests (test support)
        h = np.array([eco.splitmix64(0xABCD ^ (11 + i)) for i in range(N)], dtype=np.uint64)
        for ct, lo, hi in ((ec.UInt8, 3, 250), (ec.UInt16, 1, 65535), (ec.UInt32, 7, 4000000000)):
            w = ec.CellBuffer.empty(N + 8, ct).shard(off, N)
            ec._ffi.check(L.ec_synth_fill(ct, w.mem.ptr, N, 0xABCD, 11, float(lo), float(hi), None))
            assert np.array_equal(w.to_numpy(), (np.uint64(lo) + h % np.uint64(hi - lo + 1)).astype(ec.NP_DTYPES[ct]))
        unit = (h >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
        for ct in (ec.Float32, ec.Float64):
            w = ec.CellBuffer.empty(N + 8, ct).shard(off, N)
            ec._ffi.check(L.ec_synth_fill(ct, w.mem.ptr, N, 0xABCD, 11, -1000.0, 1000.0, None))
            assert np.array_equal(bits_of(w.to_numpy()), bits_of((-1000.0 + 2000.0 * unit).astype(ec.NP_DTYPES[ct])))
        wm = ec.Mask.empty(N + 8).shard(off, N)
        ec._ffi.check(L.ec_synth_mask(wm.mem.ptr, N, 0xABCD, 11, 30, None))
        assert np.array_equal(wm.to_numpy(), (h % np.uint64(100) >= np.uint64(30)).astype(np.uint8))
    finally:
        L.ec_tune_set(b"map_u", 2)
        L.ec_tune_set(b"unaligned_vector", 1)


_BY_WIDTH = {1: [eco.U8, eco.I8], 2: [eco.U16, eco.I16], 4: [eco.U32, eco.I32, eco.F32], 8: [eco.U64, eco.I64, eco.F64]}
_CLASSES = (0, 1, 2, 4, 8)
_SCALARS = (2.5, 3, -0.75, 7)


def _class_case(ec, pool, classes, nops, tick):
    """Operands whose load classes are `classes`: a class-c slot is a buffer of a c-byte cell type (kinds rotate with
    `tick`) at its own offset; a class-0 slot is a scalar, or — when an earlier slot is a buffer — every other time the
    same buffer as that slot.  Returns (device operands, host operands as the oracle sees them)."""
    host, dev, _, _ = pool
    ops, hs = [], []
    for k in range(nops):
        c = classes[k]
        if c:
            ct = _BY_WIDTH[c][(tick + k) % len(_BY_WIDTH[c])]
            ops.append(dev[ct].shard(k, N))
            hs.append(host[ct][k:k + N])
        else:
            earlier = [j for j in range(k) if not np.isscalar(ops[j])]
            if earlier and (tick + k) % 2 == 0:
                j = earlier[(tick // 2) % len(earlier)]
                ops.append(ops[j])
                hs.append(hs[j])
            else:
                sc = _SCALARS[(tick + k) % len(_SCALARS)]
                ops.append(sc)
                hs.append(np.full(N, float(sc)))  # a scalar operand is widened to f64 once, on the host (ec_fused.hip)
    return ops, hs


@pytest.mark.parametrize("cx", _CLASSES)
def test_fused_any_every_class_kernel(ec, pool, cx):
    """All 625 k_fused_any<CX,CY,CZ,CW> kernels (624 reachable: at least one operand is a buffer), four-operand chains
    for every class quadruple and three-operand chains for every triple, cell kinds, aliases, scalars and op triples
    rotating: against the oracle's chain under the single-op NaN rule; nothing is allocated."""
    try:
        tick = 0
        for cy in _CLASSES:
            for cz in _CLASSES:
                for cw in _CLASSES:
                    for nops in ((4, 3) if cw == 0 else (4,)):
                        tick += 1
                        classes = (cx, cy, cz, cw)
                        if not any(classes[:nops]):
                            continue  # all scalars: the ABI refuses it (checked in test_abi_host.py)
                        ops, hs = _class_case(ec, pool, classes, nops, tick)
                        if all(np.isscalar(o) for o in ops):
                            continue
                        o1, o2, o3 = OPS[tick % 4], OPS[(tick // 4) % 4], OPS[(tick // 16) % 4]
                        before = _pool_allocs(ec)
                        if nops == 4:
                            got = ec.fused.expr(ops[0], o1, ops[1], o2, ops[2], o3, ops[3])
                            eo, loose = _oracle_chain(o1, hs[0], hs[1], o2, hs[2], o3, hs[3])
                        else:
                            got = ec.fused.expr(ops[0], o1, ops[1], o2, ops[2])
                            eo, loose = _oracle_chain(o1, hs[0], hs[1], o2, hs[2])
                        assert _pool_allocs(ec) == before + 1, "one allocation: the result buffer"  # expr() allocates `out` only
                        try:
                            assert_f64_bits_equal(got.to_numpy(), eo, nan_by_class_where=loose)
                        except AssertionError as e:
                            raise AssertionError(f"classes {classes} nops {nops} ops {(o1, o2, o3)}: {e}") from None
    finally:
        pass


def test_fused_three_and_four_cell_types_with_masks(ec, pool):
    """Chains over three and four cell types (rounds 1-2: convert-then-fuse) — masked operands, odd windows, ragged
    lengths — against the oracle's chain and mask AND; no pooled temporaries."""
    host, dev, m, dm = pool
    quads = [(eco.U8, eco.U16, eco.F32, eco.F64), (eco.I64, eco.U8, eco.I16, eco.F32), (eco.F64, eco.U64, eco.I8, eco.U32),
             (eco.U16, eco.F32, eco.F64, eco.F64), (eco.I32, eco.I32, eco.U8, eco.I64)]
    for qi, q in enumerate(quads):
        for n, off in ((N, 0), (1, 0), (2, 1), (515, 1), (2049, 3)):
            bufs = [dev[t].shard(off + k, n) for k, t in enumerate(q)]
            hs = [host[t][off + k:off + k + n] for k, t in enumerate(q)]
            masks = [dm[k % 2].shard(off + k, n) for k in range(4)]
            hm = [m[k % 2][off + k:off + k + n] for k in range(4)]
            o1, o2, o3 = OPS[qi % 4], OPS[(qi + 3) % 4], OPS[(qi + 1) % 4]
            before = _pool_allocs(ec)
            got = ec.fused.expr(bufs[0], o1, bufs[1], o2, bufs[2], o3, bufs[3])
            assert _pool_allocs(ec) == before + 1  # the result buffer
            eo, loose = _oracle_chain(o1, hs[0], hs[1], o2, hs[2], o3, hs[3])
            assert_f64_bits_equal(got.to_numpy(), eo, nan_by_class_where=loose)
            mb = [ec.MaskedCellBuffer(b, k) for b, k in zip(bufs, masks)]
            gm = ec.fused.expr(mb[0], o1, mb[1], o2, mb[2])                      # three operands, three masks
            eo3, loose3 = _oracle_chain(o1, hs[0], hs[1], o2, hs[2])
            assert_f64_bits_equal(gm.buffer().to_numpy(), eo3, nan_by_class_where=loose3)
            assert np.array_equal(gm.mask().to_numpy(), hm[0] & hm[1] & hm[2])
            gm = ec.fused.expr(mb[0], o1, mb[1], o2, mb[2], o3, 2.5)             # a scalar in the last slot carries no mask
            eo4, loose4 = _oracle_chain(o1, hs[0], hs[1], o2, hs[2], o3, np.full(n, 2.5))
            assert_f64_bits_equal(gm.buffer().to_numpy(), eo4, nan_by_class_where=loose4)
            assert np.array_equal(gm.mask().to_numpy(), hm[0] & hm[1] & hm[2])


def test_every_kernel_family_with_every_load_non_temporal(ec, pool):
    """The launch's load policy (cache_plan / policy_arms): the test-sized operands of this file all fit the Infinity Cache
    budget, so every sweep above runs the default-policy arms; this one repeats the sweeps with the budget at 0 — the
    nt arm of every kernel — and, for the two-stream kernels, with a budget that admits exactly the smaller operand
    (the mixed arms)."""
    L = ec.lib()
    try:
        for budget_mb in (0,):
            L.ec_tune_set(b"mall_mb", budget_mb)
            test_masked_binop_direct_every_pair(ec, pool)
            test_map_kernels_every_instantiation(ec, pool, 2, 1, 0)
            for cx in _CLASSES:
                test_fused_any_every_class_kernel(ec, pool, cx)
            test_fused_three_and_four_cell_types_with_masks(ec, pool)
    finally:
        L.ec_tune_set(b"mall_mb", 256)
    # mixed arms: 3 MiB budget, operands of 1 / 2 / 4 / 8 MiB
    from vectors import rand_cells, rand_mask
    n = 1 << 20
    try:
        L.ec_tune_set(b"mall_mb", 3)
        host = {ct: rand_cells(ct, n, 7100 + ct) for ct in (eco.U8, eco.U16, eco.F32, eco.F64)}
        dev = {ct: ec.CellBuffer.from_vec(a) for ct, a in host.items()}
        hm = [rand_mask(n, 7200 + k) for k in range(2)]
        dm = [ec.Mask.new(x) for x in hm]
        for lt, rt in ((eco.U8, eco.U16), (eco.U16, eco.U8), (eco.F64, eco.U8), (eco.U16, eco.F32), (eco.F32, eco.F64)):
            for op in OPS:
                exp = eco.f_binop(op, host[lt], host[rt])
                assert_f64_bits_equal(dev[lt]._binop(op, dev[rt]).to_numpy(), exp, nan_by_class_where=_loose(op, host[lt], host[rt]))
            got = ec.MaskedCellBuffer(dev[lt], dm[0])._binop(eco.MUL, ec.MaskedCellBuffer(dev[rt], dm[1]))
            assert_f64_bits_equal(got.buffer().to_numpy(), eco.f_binop(eco.MUL, host[lt], host[rt]),
                                  nan_by_class_where=_loose(eco.MUL, host[lt], host[rt]))
            assert np.array_equal(got.mask().to_numpy(), hm[0] & hm[1])
        got = ec.fused.expr(dev[eco.U8], eco.SUB, dev[eco.U16], eco.DIV, dev[eco.F32], eco.ADD, dev[eco.F64])
        eo, loose = _oracle_chain(eco.SUB, host[eco.U8], host[eco.U16], eco.DIV, host[eco.F32], eco.ADD, host[eco.F64])
        assert_f64_bits_equal(got.to_numpy(), eo, nan_by_class_where=loose)
        for ct in (eco.U8, eco.U16, eco.F32):
            gm = ec.MaskedCellBuffer(dev[ct], dm[0]).min_max()
            em = eco.f_min_max(host[ct], hm[0])
            assert (gm[0].bits(), gm[1].bits()) == (em[0].bits(), em[1].bits())
            sel = ec.MaskedCellBuffer(dev[ct], dm[1]).to_vec_with_nodata(ct, ec.NoData.default())
            assert np.array_equal(bits_of(sel), bits_of(eco.f_mask_select(host[ct], hm[1], eco.nodata_value(eco.ND_DEFAULT, ct))))
    finally:
        L.ec_tune_set(b"mall_mb", 256)


def test_scalar_kernels_with_and_without_the_nan_rule(ec, pool):
    """buffer ∘ scalar runs without cv_bin_op!'s NaN rule when no result can be a NaN — integer cells, a finite scalar, non-zero for a
    divide — and with it otherwise.  Both forms, every integer and float cell type, scalars on both sides of every condition (0, -0, ±inf,
    NaN, a finite one, one that overflows the product), against the oracle bit for bit (a NaN scalar against a NaN cell: by class, the
    reference leaves the winner to the compiler for + and *)."""
    host, dev, m, dm = pool
    scalars = [2.5, -3.0, 0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1e308, 5e-324]
    for ct in range(NT):
        h = host[ct][3:3 + N]
        d = dev[ct].shard(3, N)
        for sc in scalars:
            for op in OPS:
                exp = eco.f_binop(op, h, np.full(N, sc))
                got = d._binop(op, sc).to_numpy()
                try:
                    assert_f64_bits_equal(got, exp, nan_by_class_where=_loose(op, h, np.full(N, sc)))
                except AssertionError as e:
                    raise AssertionError(f"cell type {ct} op {op} scalar {sc!r}: {e}") from None


def test_every_load_policy_arm_by_force(ec, pool):
    """`cache_force` pins the launch's load policy to given bits (the A/B hook behind profiles/r04/cache_plan_ab.md): every arm of the
    two-stream kernels (00, 01, 10, 11) and a spread of the four-stream masked kernel's sixteen computes the oracle's cells — the policy
    changes instructions' cache modifiers, never results; and the equal-peer rule's answer for the launch shapes it was made for is
    visible through the hook's absence (results again)."""
    host, dev, m, dm = pool
    L = ec.lib()
    try:
        for force in (0, 1, 2, 3):
            L.ec_tune_set(b"cache_force", force)
            for lt, rt in ((eco.U8, eco.U16), (eco.U8, eco.U8), (eco.I16, eco.F32), (eco.F64, eco.I64)):
                for op in OPS:
                    exp = eco.f_binop(op, host[lt][:N], host[rt][:N])
                    got = dev[lt].shard(0, N)._binop(op, dev[rt].shard(0, N))
                    assert_f64_bits_equal(got.to_numpy(), exp, nan_by_class_where=_loose(op, host[lt][:N], host[rt][:N]))
                got = dev[lt].shard(1, N)._binop(eco.MUL, 2.5)
                assert_f64_bits_equal(got.to_numpy(), eco.f_binop(eco.MUL, host[lt][1:1 + N], np.full(N, 2.5)))
        for force in (0, 5, 10, 15, 6, 9):
            L.ec_tune_set(b"cache_force", force)
            a, b = ec.MaskedCellBuffer(dev[eco.F32].shard(0, N), dm[0].shard(0, N)), ec.MaskedCellBuffer(dev[eco.U8].shard(0, N), dm[1].shard(0, N))
            got = a._binop(eco.DIV, b)
            assert_f64_bits_equal(got.buffer().to_numpy(), eco.f_binop(eco.DIV, host[eco.F32][:N], host[eco.U8][:N]),
                                  nan_by_class_where=_loose(eco.DIV, host[eco.F32][:N], host[eco.U8][:N]))
            assert np.array_equal(got.mask().to_numpy(), m[0][:N] & m[1][:N])
    finally:
        L.ec_tune_set(b"cache_force", -1)


# ---------------------------------------------------------------- expression programs (ec_expr): trees of any depth in one pass
def _oracle_program(streams, scalars, steps):
    """The program evaluated step by step on the oracle's typed loops — the eager chain the reference would run — and the
    cells whose NaN bits the reference leaves open (both operands NaN under a commutative op, carried forward)."""
    n = min(len(s_) for s_ in streams)
    val, loose = {}, {}
    for k, s_ in enumerate(streams):
        val[k], loose[k] = s_[:n], np.zeros(n, bool)
    for k, c in enumerate(scalars):
        val[8 + k], loose[8 + k] = np.full(n, float(c)), np.zeros(n, bool)  # a scalar is widened to f64 once, on the host
    last = None
    for op, a, b, dst in steps:
        r = eco.f_binop(op, val[a], val[b])
        lo = loose[a] | loose[b]
        if op in (eco.ADD, eco.MUL):
            with np.errstate(all="ignore"):
                lo = lo | (np.isnan(val[a].astype(np.float64)) & np.isnan(val[b].astype(np.float64)))
        val[4 + dst], loose[4 + dst] = r, lo
        last = 4 + dst
    return val[last], loose[last]


def test_expr_every_stream_class_kernel(ec, pool):
    """All 340 k_expr<C0,C1,C2,C3> kernels (stream byte widths, packed): a program that uses every stream, a scalar and
    three registers, cell kinds and ops rotating — against the oracle's step-by-step evaluation; one allocation (the
    result) per call."""
    host, dev, _, _ = pool
    P = ec.fused
    tick = 0
    for ns in (1, 2, 3, 4):
        import itertools
        for classes in itertools.product((1, 2, 4, 8), repeat=ns):
            tick += 1
            cts = [_BY_WIDTH[c][(tick + k) % len(_BY_WIDTH[c])] for k, c in enumerate(classes)]
            bufs = [dev[ct].shard(k, N) for k, ct in enumerate(cts)]
            hs = [host[ct][k:k + N] for k, ct in enumerate(cts)]
            ops = [OPS[(tick + j) % 4] for j in range(6)]
            steps = [(ops[0], P.STREAM0, P.SCALAR0, 0)]                                  # r0 = s0 op 2.5
            for k in range(1, ns):
                steps.append((ops[k], P.REG0 + (k - 1) % 3, P.STREAM0 + k, k % 3))     # r[k] = r[k-1] op s_k   (registers 0..2)
            steps.append((ops[4], P.STREAM0 + ns - 1, P.REG0 + (ns - 1) % 3, 3))        # r3 = s_last op r
            steps.append((ops[5], P.REG0 + 3, P.SCALAR0 + 1, (ns - 1) % 3))             # r = r3 op 7
            before = _pool_allocs(ec)
            got = P.program(bufs, [2.5, 7], steps)
            assert _pool_allocs(ec) == before + 1
            eo, loose = _oracle_program(hs, [2.5, 7], steps)
            try:
                assert_f64_bits_equal(got.to_numpy(), eo, nan_by_class_where=loose)
            except AssertionError as e:
                raise AssertionError(f"classes {classes} types {cts} steps {steps}: {e}") from None


def test_expr_random_programs_masks_windows_and_the_eager_chain(ec, pool):
    """Random valid programs (1-4 streams of any types, 0-8 scalars, 1-16 steps, registers reused) over ragged lengths and
    odd windows: against the oracle, and — where every scalar stands on the right, as the reference's operators require —
    against the same operators run one by one through the eager HIP path; masked programs AND the streams' masks."""
    host, dev, m, dm = pool
    P = ec.fused
    rng = np.random.default_rng(int(os.environ.get("EC_EXPR_SEED", "4321")))  # EC_EXPR_SEED / EC_EXPR_TRIALS: longer one-off hunts
    for trial in range(int(os.environ.get("EC_EXPR_TRIALS", "60"))):
        ns = int(rng.integers(1, 5))
        n, off = [(N, 0), (1, 0), (2, 1), (515, 1), (2049, 3)][trial % 5]
        cts = [int(c) for c in rng.integers(0, NT, size=ns)]
        bufs = [dev[ct].shard(off + k, n) for k, ct in enumerate(cts)]
        hs = [host[ct][off + k:off + k + n] for k, ct in enumerate(cts)]
        scalars = [float(x) for x in rng.choice([2.5, -3.0, 0.5, 7.0, 1e-3, 65536.0, -0.0, 1.0], size=int(rng.integers(0, 9)))]
        written, steps = [], []
        for _ in range(int(rng.integers(1, 17))):
            def ref(left):
                kinds = ["s"] + (["r"] if written else []) + (["c"] if scalars and not left else [])
                kind = kinds[int(rng.integers(0, len(kinds)))]
                if kind == "s":
                    return P.STREAM0 + int(rng.integers(0, ns))
                if kind == "r":
                    return P.REG0 + int(rng.choice(written))
                return P.SCALAR0 + int(rng.integers(0, len(scalars)))
            dst = int(rng.integers(0, 4))
            steps.append((int(rng.integers(0, 4)), ref(True), ref(False), dst))
            if dst not in written:
                written.append(dst)
        got = P.program(bufs, scalars, steps)
        eo, loose = _oracle_program(hs, scalars, steps)
        assert_f64_bits_equal(got.to_numpy(), eo, nan_by_class_where=loose)
        if trial % 4 == 1:  # k_expr_cellwise: the any-alignment path behind the unaligned_vector knob
            ec.lib().ec_tune_set(b"unaligned_vector", 0)
            try:
                cw = (P.program([ec.MaskedCellBuffer(b, dm[k % 2].shard(off + k, n)) for k, b in enumerate(bufs)], scalars, steps)
                      if trial % 8 == 1 else P.program(bufs, scalars, steps))
            finally:
                ec.lib().ec_tune_set(b"unaligned_vector", 1)
            assert_f64_bits_equal((cw.buffer() if trial % 8 == 1 else cw).to_numpy(), eo, nan_by_class_where=loose)
        # the eager chain on the device: every step one operator call
        regs = {}
        for op, a, b, dst in steps:
            x = bufs[a] if a < 4 else regs[a - 4]
            y = bufs[b] if b < 4 else regs[b - 4] if b < 8 else scalars[b - 8]
            regs[dst] = x._binop(op, y)
        eager = regs[steps[-1][3]].to_numpy()
        assert_f64_bits_equal(got.to_numpy(), eager, nan_by_class_where=loose)
        if trial % 3 == 0:  # masked: the result's mask is the AND of every stream's mask (used by the program or not)
            mb = [ec.MaskedCellBuffer(b, dm[k % 2].shard(off + k, n)) for k, b in enumerate(bufs)]
            gm = P.program(mb, scalars, steps)
            assert_f64_bits_equal(gm.buffer().to_numpy(), eo, nan_by_class_where=loose)
            want = np.ones(n, np.uint8)
            for k in range(ns):
                want &= m[k % 2][off + k:off + k + n]
            assert np.array_equal(gm.mask().to_numpy(), want)


def test_lazy_trees_deeper_than_two_levels_run_as_one_pass(ec, pool):
    """`lazy()` operator syntax: EVI over three u16 bands (8 operators, 4 scalars) and a right-heavy tree are ONE launch
    (one allocation: the result), bit-identical to the eager evaluation operator by operator; a tree that needs more than
    four buffers is cut and still correct."""
    host, dev, _, _ = pool
    L = ec.fused.lazy
    nir, red, blue = dev[eco.U16].shard(0, N), dev[eco.U16].shard(3, N), dev[eco.U16].shard(5, N)
    before = _pool_allocs(ec)
    evi = (((L(nir) - red) * 2.5) / (((L(nir) + L(red) * 6.0) - L(blue) * 7.5) + 1.0)).eval()
    assert _pool_allocs(ec) == before + 1, "EVI must be one pass: no temporaries"
    eager = ((nir - red) * 2.5) / (((nir + red * 6.0) - blue * 7.5) + 1.0)
    assert np.array_equal(bits_of(evi.to_numpy()), bits_of(eager.to_numpy()))
    hn, hr, hb = host[eco.U16][:N], host[eco.U16][3:3 + N], host[eco.U16][5:5 + N]
    f = lambda op, a, b: eco.f_binop(op, a, b if isinstance(b, np.ndarray) else np.full(N, float(b)))  # noqa: E731
    exp = f(eco.DIV, f(eco.MUL, f(eco.SUB, hn, hr), 2.5),
            f(eco.ADD, f(eco.SUB, f(eco.ADD, hn, f(eco.MUL, hr, 6.0)), f(eco.MUL, hb, 7.5)), 1.0))
    assert_f64_bits_equal(evi.to_numpy(), exp)  # integer bands: no NaN can arise before the divide, none is left open
    # the way a formula is usually written, scalars on the left (README): the same cells (products commute exactly)
    evi_l = (2.5 * (L(nir) - red) / (L(nir) + 6.0 * L(red) - 7.5 * L(blue) + 1.0)).eval()
    assert np.array_equal(bits_of(evi_l.to_numpy()), bits_of(evi.to_numpy()))
    # right-heavy two-operator tree: not a shape of the two-level kernel, one pass through the program kernel
    a, b, c = dev[eco.F32].shard(0, N), dev[eco.I16].shard(1, N), dev[eco.U8].shard(2, N)
    before = _pool_allocs(ec)
    got = (L(a) - (L(b) * c)).eval()
    assert _pool_allocs(ec) == before + 1
    assert np.array_equal(bits_of(got.to_numpy()), bits_of((a - (b * c)).to_numpy()))
    # five distinct buffers: does not fit the four streams -> cut into pieces, same bits
    d, e = dev[eco.F64].shard(4, N), dev[eco.U32].shard(6, N)
    got = (((L(a) + b) * c) - ((L(d) / e) + a)).eval()
    assert np.array_equal(bits_of(got.to_numpy()), bits_of((((a + b) * c) - ((d / e) + a)).to_numpy()))


def test_lazy_random_trees_against_the_oracle(ec, pool):
    """Random operator trees (depth ≤ 5, leaves from six buffers of different types and scalars on either side) written
    with `lazy()`: whatever the evaluator does with them — the two-level kernel, one expression program, or a cut into
    pieces when the tree needs more than four buffers or registers — the cells are the oracle's, evaluated node by node."""
    host, dev, _, _ = pool
    L = ec.fused.lazy
    rng = np.random.default_rng(99)
    cts = [eco.U16, eco.F32, eco.U8, eco.I32, eco.F64, eco.I16]
    bufs = [dev[ct].shard(k, N) for k, ct in enumerate(cts)]
    hs = [host[ct][k:k + N] for k, ct in enumerate(cts)]

    def grow(depth, nbuf):
        """-> (lazy tree or scalar, oracle value, loose, has_buffer)"""
        if depth == 0 or rng.random() < 0.25:
            if rng.random() < 0.3:
                c = float(rng.choice([2.5, -3.0, 0.5, 7.0, 65536.0, 1.0]))
                return c, np.full(N, c), np.zeros(N, bool), False
            k = int(rng.integers(0, nbuf))
            return L(bufs[k]), hs[k], np.zeros(N, bool), True
        op = int(rng.integers(0, 4))
        a, va, la, ba = grow(depth - 1, nbuf)
        b, vb, lb, bb = grow(depth - 1, nbuf)
        if not (ba or bb):  # two scalars: keep one side a buffer so that the node is an operator of the library
            a, va, la, ba = L(bufs[0]), hs[0], np.zeros(N, bool), True
        t = [a + b, a - b, a * b, a / b][op] if not isinstance(a, float) or not isinstance(b, float) else None
        lo = la | lb
        if op in (eco.ADD, eco.MUL):
            lo = lo | _both_nan(va, vb)
        return t, eco.f_binop(op, va, vb), lo, True

    seen_programs = 0
    for trial in range(40):
        t, exp, loose, _ = grow(int(rng.integers(2, 6)), 3 if trial % 2 else 6)
        if not isinstance(t, ec.fused.Lazy) or t.op is None:
            continue
        before = _pool_allocs(ec)
        got = t.eval()
        seen_programs += _pool_allocs(ec) == before + 1 and t._depth() > 2
        assert_f64_bits_equal(got.to_numpy(), exp, nan_by_class_where=loose)
    assert seen_programs >= 5, "deep trees over three buffers should have run as single programs"

    # the same with masked buffers: the result's mask is the AND of the masks of the buffers the tree uses
    _, _, m, dm = pool
    mbufs = [ec.MaskedCellBuffer(b, dm[k % 2].shard(k, N)) for k, b in enumerate(bufs)]
    hmask = [m[k % 2][k:k + N].astype(bool) for k in range(len(bufs))]

    def grow_masked(depth, nbuf):
        if depth == 0 or rng.random() < 0.25:
            if rng.random() < 0.3:
                c = float(rng.choice([2.5, -3.0, 0.5, 7.0]))
                return c, np.full(N, c), np.zeros(N, bool), np.ones(N, bool), False
            k = int(rng.integers(0, nbuf))
            return L(mbufs[k]), hs[k], np.zeros(N, bool), hmask[k], True
        op = int(rng.integers(0, 4))
        a, va, la, ma, ba = grow_masked(depth - 1, nbuf)
        b, vb, lb, mb, bb = grow_masked(depth - 1, nbuf)
        if not (ba or bb):
            a, va, la, ma, ba = L(mbufs[0]), hs[0], np.zeros(N, bool), hmask[0], True
        lo = la | lb
        if op in (eco.ADD, eco.MUL):
            lo = lo | _both_nan(va, vb)
        return [a + b, a - b, a * b, a / b][op], eco.f_binop(op, va, vb), lo, ma & mb, True

    for trial in range(12):
        t, exp, loose, valid, _ = grow_masked(int(rng.integers(2, 5)), 3 if trial % 2 else 6)
        if not isinstance(t, ec.fused.Lazy) or t.op is None:
            continue
        got = t.eval()
        assert_f64_bits_equal(got.buffer().to_numpy(), exp, nan_by_class_where=loose)
        assert np.array_equal(got.mask().to_numpy().astype(bool), valid)


def _stat(ec, key):
    v = C.c_int64(0)
    assert ec.lib().ec_stat_get(key, C.byref(v)) == 0
    return v.value


def test_expr_compiled_form_equals_the_interpreter_and_the_oracle(ec, pool):
    """The run-time compiled form of a program (`expr_jit` = 2: hiprtc on the calling thread): every cell type as a stream,
    random programs, odd windows (peeled head, odd tail), masks — bit-identical to the interpreter (`expr_jit` = 0) and to
    the oracle evaluated step by step; the counters say which path ran; a program first met inside a stream capture is
    interpreted there."""
    host, dev, m, dm = pool
    P, L = ec.fused, ec.lib()
    rng = np.random.default_rng(int(os.environ.get("EC_EXPR_SEED", "777")))

    def both(bufs, scalars, steps):
        L.ec_tune_set(b"expr_jit", 2)
        try:
            j0, i0 = _stat(ec, b"expr_jit_launches"), _stat(ec, b"expr_interp_launches")
            jit = P.program(bufs, scalars, steps)
            assert _stat(ec, b"expr_jit_launches") == j0 + 1 and _stat(ec, b"expr_interp_launches") == i0
            L.ec_tune_set(b"expr_jit", 0)
            interp = P.program(bufs, scalars, steps)
            assert _stat(ec, b"expr_interp_launches") == i0 + 1
        finally:
            L.ec_tune_set(b"expr_jit", 1)
        return jit, interp

    # every cell type as stream 0 (typed loads of the generated kernel), with a second stream of another width
    for ct in range(NT):
        other = [eco.F32, eco.U8, eco.I64, eco.U16][ct % 4]
        for n, off in ((N, 0), (2049, 3), (1, 0), (2, 1)):
            bufs = [dev[ct].shard(off, n), dev[other].shard(off + 1, n)]
            hs = [host[ct][off:off + n], host[other][off + 1:off + 1 + n]]
            steps = [(OPS[ct % 4], P.STREAM0, P.STREAM0 + 1, 0), (eco.MUL, P.REG0, P.SCALAR0, 1), (OPS[(ct + 1) % 4], P.STREAM0 + 1, P.REG0 + 1, 0),
                     (eco.DIV, P.REG0, P.STREAM0, 2)]
            jit, interp = both(bufs, [2.5], steps)
            eo, loose = _oracle_program(hs, [2.5], steps)
            assert np.array_equal(bits_of(jit.to_numpy()), bits_of(interp.to_numpy())), (ct, n, off)
            assert_f64_bits_equal(jit.to_numpy(), eo, nan_by_class_where=loose)
    compiles = _stat(ec, b"expr_jit_compiles")
    assert compiles >= NT and _stat(ec, b"expr_jit_failures") == 0
    # random programs, masked every third
    for trial in range(int(os.environ.get("EC_EXPR_JIT_TRIALS", "24"))):
        ns = int(rng.integers(1, 5))
        n, off = [(N, 0), (515, 1), (2049, 3)][trial % 3]
        cts = [int(c) for c in rng.integers(0, NT, size=ns)]
        bufs = [dev[ct].shard(off + k, n) for k, ct in enumerate(cts)]
        hs = [host[ct][off + k:off + k + n] for k, ct in enumerate(cts)]
        scalars = [float(x) for x in rng.choice([2.5, -3.0, 0.5, 7.0, 1e-3, -0.0], size=int(rng.integers(1, 9)))]
        written, steps = [], []
        for _ in range(int(rng.integers(1, 17))):
            def ref():
                kind = ["s", "c"] + (["r"] if written else [])
                kind = kind[int(rng.integers(0, len(kind)))]
                return (P.STREAM0 + int(rng.integers(0, ns)) if kind == "s" else P.SCALAR0 + int(rng.integers(0, len(scalars))) if kind == "c"
                        else P.REG0 + int(rng.choice(written)))
            dst = int(rng.integers(0, 4))
            steps.append((int(rng.integers(0, 4)), ref(), ref(), dst))
            if dst not in written:
                written.append(dst)
        if trial % 3 == 0:
            bufs = [ec.MaskedCellBuffer(b, dm[k % 2].shard(off + k, n)) for k, b in enumerate(bufs)]
        jit, interp = both(bufs, scalars, steps)
        eo, loose = _oracle_program(hs, scalars, steps)
        jv, iv = (jit.buffer(), interp.buffer()) if trial % 3 == 0 else (jit, interp)
        assert np.array_equal(bits_of(jv.to_numpy()), bits_of(iv.to_numpy())), (trial, cts, steps)
        assert_f64_bits_equal(jv.to_numpy(), eo, nan_by_class_where=loose)
        if trial % 3 == 0:
            assert np.array_equal(jit.mask().to_numpy(), interp.mask().to_numpy())
    # the same program again: served from the cache, nothing compiled
    before = _stat(ec, b"expr_jit_compiles")
    both(bufs, scalars, steps)
    assert _stat(ec, b"expr_jit_compiles") == before
    # a program first met inside a stream capture: no compile-and-load there — interpreted, and correct on replay
    import torch
    x = dev[eco.U16].shard(0, N)
    out = ec.CellBuffer.empty(N, ec.Float64)
    E = ec._ffi
    dt, p = (C.c_uint8 * 1)(eco.U16), (C.c_void_p * 1)(x.mem.ptr)
    sc = (E.EcValue * 1)(ec.CellValue.new(123.25).to_ec())
    st = (E.EcExprStep * 2)(E.EcExprStep(eco.MUL, 0, 8, 3), E.EcExprStep(eco.SUB, 7, 0, 1))
    cap = torch.cuda.Stream()
    L.ec_tune_set(b"expr_jit", 2)
    try:
        chk = E.check
        chk(L.ec_prepare_stream(cap.cuda_stream))
        i0 = _stat(ec, b"expr_interp_launches")
        g = torch.cuda.CUDAGraph()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=cap):
            chk(L.ec_expr(dt, p, 1, sc, 1, st, 2, N, out.mem.ptr, cap.cuda_stream))
        g.replay()
        torch.cuda.synchronize()
        assert _stat(ec, b"expr_interp_launches") == i0 + 1, "a program first met inside a capture must be interpreted"
        h = host[eco.U16][:N]
        exp = eco.f_binop(eco.SUB, eco.f_binop(eco.MUL, h, np.full(N, 123.25)), h)
        assert_f64_bits_equal(out.to_numpy(), exp)
        # once its module is loaded (one launch outside a capture), the compiled kernel is what a capture records
        chk(L.ec_expr(dt, p, 1, sc, 1, st, 2, N, out.mem.ptr, cap.cuda_stream))
        cap.synchronize()
        j0 = _stat(ec, b"expr_jit_launches")
        out2 = ec.CellBuffer.empty(N, ec.Float64)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=cap):
            chk(L.ec_expr(dt, p, 1, sc, 1, st, 2, N, out2.mem.ptr, cap.cuda_stream))
        assert _stat(ec, b"expr_jit_launches") == j0 + 1
        g2.replay()
        torch.cuda.synchronize()
        assert_f64_bits_equal(out2.to_numpy(), exp)
    finally:
        L.ec_tune_set(b"expr_jit", 1)


def test_expr_background_compile_takes_over_once_a_program_has_run_long_enough(ec):
    """`expr_jit` = 1 (the default): launches interpret until a program has interpreted 2^31 cell-steps, a background thread
    compiles it, and later launches run the compiled form — same cells before and after."""
    import time
    P, L, E = ec.fused, ec.lib(), ec._ffi
    n = 1 << 24
    x = ec.CellBuffer.empty(n, ec.UInt16)
    E.check(L.ec_synth_fill(ec.UInt16, x.mem.ptr, n, 0xABCDEF, 0, 1.0, 60000.0, ec.stream()))
    steps = [(eco.MUL, P.STREAM0, P.SCALAR0, 0)] + [(OPS[k % 3], P.REG0, P.SCALAR0 + (k % 2), 0) for k in range(15)]  # 16 steps
    scalars = [1.0009765625, 3.0]
    c0, f0 = _stat(ec, b"expr_jit_compiles"), _stat(ec, b"expr_jit_failures")
    first = P.program([x], scalars, steps).to_numpy()
    j0 = _stat(ec, b"expr_jit_launches")
    for _ in range(7):  # 8 launches x 2^24 cells x 16 steps = 2^31: the eighth (itself interpreted) hands the program to the compiler
        P.program([x], scalars, steps)
    assert _stat(ec, b"expr_jit_launches") == j0, "all eight launches were interpreted"
    deadline = time.time() + 60
    while _stat(ec, b"expr_jit_compiles") == c0 and _stat(ec, b"expr_jit_failures") == f0 and time.time() < deadline:
        time.sleep(0.05)
    assert _stat(ec, b"expr_jit_failures") == f0 and _stat(ec, b"expr_jit_compiles") == c0 + 1
    later = P.program([x], scalars, steps)
    assert _stat(ec, b"expr_jit_launches") == j0 + 1
    assert np.array_equal(bits_of(later.to_numpy()), bits_of(first))


def test_host_to_host_expression_pipeline(ec):
    """`ec_host_expr`: host arrays in, host array out, chunked with upload / kernel / download overlapped — pageable numpy
    arrays (registered for the call), page-locked ones (`pinned_empty`), chunk sizes that do and do not divide the length,
    one chunk, odd lengths; against the oracle's step-by-step evaluation; a malformed program is refused up front."""
    P = ec.fused
    n = (1 << 20) + 12345
    hs = [rand_cells(ct, n, 9100 + ct) for ct in (eco.U16, eco.I8, eco.F32)]
    S, R, K = (lambda k: k), (lambda k: 4 + k), (lambda k: 8 + k)
    steps = [(eco.SUB, S(0), S(1), 0), (eco.MUL, R(0), K(0), 0), (eco.ADD, S(2), R(0), 1), (eco.DIV, R(1), S(0), 0)]
    exp, loose = _oracle_program(hs, [2.5], steps)
    for chunk in (0, 1 << 18, 300001, n, 7):
        if chunk == 7:  # many tiny chunks: only a prefix, or the test would take minutes
            got = P.program_host([h[:1001] for h in hs], [2.5], steps, chunk_cells=7)
            assert_f64_bits_equal(got, exp[:1001], nan_by_class_where=loose[:1001])
            continue
        got = P.program_host(hs, [2.5], steps, chunk_cells=chunk)
        assert_f64_bits_equal(got, exp, nan_by_class_where=loose)
    # page-locked operands and result
    pinned = [P.pinned_empty(n, h.dtype) for h in hs]
    for q, h in zip(pinned, hs):
        q[:] = h
    out = P.pinned_empty(n, np.float64)
    out[:] = -1.0
    got = P.program_host(pinned, [2.5], steps, out=out, chunk_cells=1 << 18)
    assert got.ctypes.data == out.ctypes.data
    assert_f64_bits_equal(out, exp, nan_by_class_where=loose)
    # four host threads, each its own pipeline (own streams, own staging from the shared pool), at once
    import threading
    results, errors = [None] * 4, []

    def worker(k):
        try:
            results[k] = P.program_host(hs, [2.5], steps, chunk_cells=50000 + 1111 * k)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for r in results:
        assert_f64_bits_equal(r, exp, nan_by_class_where=loose)
    # the same four threads over SHIFTED windows of the same arrays: their page-lock requests overlap only partly, so a thread waits
    # for the registration of another call to go away before it makes its own (PinSet, ec_hostpipe.hpp) — no deadlock, same cells
    big = [rand_cells(ct, n + 4000, 9200 + ct) for ct in (eco.U16, eco.I8, eco.F32)]
    results, errors = [None] * 4, []

    def shifted(k):
        try:
            results[k] = P.program_host([b[1000 * k:1000 * k + n] for b in big], [2.5], steps, chunk_cells=1 << 18)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=shifted, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k, r in enumerate(results):
        e_k, l_k = _oracle_program([b[1000 * k:1000 * k + n] for b in big], [2.5], steps)
        assert_f64_bits_equal(r, e_k, nan_by_class_where=l_k)
    # two overlapping windows of ONE host array as two operands: their pages are merged into one registration
    base = rand_cells(eco.U16, 400000, 77)
    w0, w1 = base[:300000], base[1000:301000]
    prog = [(eco.SUB, S(0), S(1), 0), (eco.MUL, R(0), K(0), 0)]
    e2, l2 = _oracle_program([w0, w1], [0.5], prog)
    assert_f64_bits_equal(P.program_host([w0, w1], [0.5], prog, chunk_cells=65536), e2, nan_by_class_where=l2)
    # in place: the result array is one of the (f64) operands
    x = np.linspace(-5.0, 5.0, 200001)
    want = eco.f_binop(eco.MUL, x, np.full(x.size, 2.5))
    got = P.program_host([x], [2.5], [(eco.MUL, S(0), K(0), 0)], out=x, chunk_cells=30000)
    assert got.ctypes.data == x.ctypes.data and np.array_equal(bits_of(x), bits_of(want))
    # a single operator is a one-step program: the reference's quick example, host to host
    q = P.program_host([np.array([1, 2, 3], np.uint8), np.array([2, 4, 6], np.uint16)], [0.5], [(eco.DIV, S(0), S(1), 0), (eco.MUL, R(0), K(0), 0)])
    assert q.tolist() == [0.25, 0.25, 0.25]
    with pytest.raises(Exception):
        P.program_host(hs, [2.5], [(eco.ADD, S(0), R(1), 0)])  # register read before it is written


def test_host_to_host_masked_expression_pipeline(ec):
    """`ec_host_masked_expr`: from_vec_with_nodata of every stream, the program with the AND of the masks, to_vec_with_nodata of
    the f64 result — in one streamed call.  Streams with and without a nodata value, integer and float nodata, ragged chunks;
    values and mask against the oracle."""
    P = ec.fused
    n = 300001
    rng = np.random.default_rng(5)
    a = rng.integers(0, 2000, n).astype(np.uint16)
    b = rng.uniform(-50, 50, n).astype(np.float32)
    c = rng.integers(-100, 100, n).astype(np.int8)
    a[rng.integers(0, n, 5000)] = 0              # nodata 0 in the u16 band
    b[rng.integers(0, n, 7000)] = np.float32(-9999.0)
    S, R, K = (lambda k: k), (lambda k: 4 + k), (lambda k: 8 + k)
    steps = [(eco.SUB, S(0), S(1), 0), (eco.MUL, S(2), K(0), 1), (eco.DIV, R(0), R(1), 2), (eco.ADD, R(2), S(0), 0)]
    vals, loose = _oracle_program([a, b, c], [0.5], steps)
    valid = (a != 0) & (b != np.float32(-9999.0))   # the i8 stream has no nodata value: all valid
    for chunk in (0, 65536, 77777):
        out, mask = P.program_host_masked([a, b, c], [0, -9999.0, None], [0.5], steps, out_nodata=-1e30, want_mask=True, chunk_cells=chunk)
        assert np.array_equal(mask, valid)
        assert_f64_bits_equal(out, np.where(valid, vals, -1e30), nan_by_class_where=loose & valid)
    # without an output nodata value the values of ALL cells come back, as the reference computes them (masked_buffer.rs:326-335)
    out = P.program_host_masked([a, b, c], [0, -9999.0, None], [0.5], steps)
    assert_f64_bits_equal(out, vals, nan_by_class_where=loose)
    # the device-resident masked path says the same
    ma = ec.MaskedCellBuffer.from_vec_with_nodata(a, ec.NoData.new(np.uint16(0)))
    mb = ec.MaskedCellBuffer.from_vec_with_nodata(b, ec.NoData.new(np.float32(-9999.0)))
    mc = ec.MaskedCellBuffer.from_vec(c)
    res = P.program([ma, mb, mc], [0.5], steps)
    assert np.array_equal(res.mask().to_numpy().astype(bool), valid)
    assert np.array_equal(bits_of(res.buffer().to_numpy()), bits_of(out))
    with pytest.raises(Exception):
        P.program_host_masked([a, b], [0, None], [], [(eco.ADD, S(0), R(0), 0)])  # malformed program: refused before any transfer


def test_expr_without_hiprtc_keeps_interpreting_and_says_so(tmp_path):
    """A machine without libhiprtc (EC_HIPRTC_LIB pointing nowhere, in a process of its own): `expr_jit` = 1 counts the failed
    compile and keeps serving the program with the interpreter; `expr_jit` = 2 refuses loudly with the loader's message."""
    import subprocess
    import sys
    code = r"""
import sys, time, ctypes as C
sys.path.insert(0, sys.argv[1])
import numpy as np
import erased_cells_hip as ec
ec.init(0)
L, P = ec.lib(), ec.fused
def stat(k):
    v = C.c_int64(0); assert L.ec_stat_get(k, C.byref(v)) == 0; return v.value
n = 1 << 24
x = ec.CellBuffer.from_vec(np.arange(n, dtype=np.uint16))
steps = [(ec.MUL, 0, 8, 0)] + [(ec.ADD, 4, 8, 0)] * 15
for _ in range(9):
    out = P.program([x], [1.5], steps)
deadline = time.time() + 30
while stat(b"expr_jit_failures") == 0 and time.time() < deadline:
    time.sleep(0.05)
assert stat(b"expr_jit_failures") == 1 and stat(b"expr_jit_compiles") == 0 and stat(b"expr_jit_launches") == 0
out = P.program([x], [1.5], steps).to_numpy()
assert out[3] == 3 * 1.5 + 15 * 1.5 and stat(b"expr_interp_launches") == 10
L.ec_tune_set(b"expr_jit", 2)
try:
    P.program([x], [1.5], [(ec.MUL, 0, 8, 0)])
    print("NOT REFUSED")
except Exception as e:
    print("REFUSED:", e)
"""
    env = dict(os.environ, EC_HIPRTC_LIB="/nonexistent/libhiprtc.so")
    r = subprocess.run([sys.executable, "-c", code, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "erased-cells_amd", "python")],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "REFUSED:" in r.stdout and "libhiprtc" in r.stdout, r.stdout


def test_expr_min_max_without_the_raster(ec, pool):
    """`ec_expr_min_max`: (min, max) of a program's result equals `min_max()` of the materialised result — bits and cell type —
    for plain and masked streams, through the two-pass form (expr_jit = 0) and the compiled reduce kernel (expr_jit = 2),
    odd windows, results that contain NaN (0/0: the x86 default NaN is negative, so it is the minimum under total_cmp),
    nothing valid at all (the fold's identities)."""
    host, dev, m, dm = pool
    P, L = ec.fused, ec.lib()
    S, R, K = (lambda k: k), (lambda k: 4 + k), (lambda k: 8 + k)
    ndvi = [(eco.SUB, S(0), S(1), 0), (eco.ADD, S(0), S(1), 1), (eco.DIV, R(0), R(1), 0)]
    cases = [([eco.U16, eco.U16], ndvi, []), ([eco.U8, eco.I8], ndvi, []),  # u8 / i8 bands hold zeros: 0/0 cells
             ([eco.F32, eco.I16, eco.F64], [(eco.MUL, S(0), K(0), 0), (eco.SUB, R(0), S(1), 1), (eco.DIV, R(1), S(2), 0), (eco.ADD, R(0), K(1), 2)], [2.5, -7.0]),
             ([eco.I64], [(eco.MUL, S(0), K(0), 0)], [0.5])]
    for cts, steps, scalars in cases:
        for n, off in ((N, 0), (2049, 3), (1, 0), (2, 1)):
            for masked in (False, True):
                bufs = [dev[ct].shard(off + k, n) for k, ct in enumerate(cts)]
                if masked:
                    bufs = [ec.MaskedCellBuffer(b, dm[k % 2].shard(off + k, n)) for k, b in enumerate(bufs)]
                want = P.program(bufs, scalars, steps).min_max()
                for mode in (0, 2):
                    with P.jit(mode):
                        j0 = _stat(ec, b"expr_jit_launches")
                        got = P.program_min_max(bufs, scalars, steps)
                        assert (_stat(ec, b"expr_jit_launches") - j0 == 1) == (mode == 2)
                    assert (got[0].ct, got[1].ct) == (ec.Float64, ec.Float64)
                    assert (got[0].bits(), got[1].bits()) == (want[0].bits(), want[1].bits()), (cts, n, off, masked, mode, got, want)
    # nothing valid: the identities of the fold, (f64::MAX, f64::MIN)
    x = ec.MaskedCellBuffer(dev[eco.U16].shard(0, N), ec.Mask.fill(N, False))
    for mode in (0, 2):
        with P.jit(mode):
            mn, mx = P.program_min_max([x], [2.0], [(eco.MUL, S(0), K(0), 0)])
        assert float(mn.value) == np.finfo(np.float64).max and float(mx.value) == np.finfo(np.float64).min
    # operator syntax: lazy(...).min_max() is the same call when the tree fits one program
    a_, b_ = dev[eco.U16].shard(0, N), dev[eco.I16].shard(2, N)
    t = (P.lazy(a_) - b_) * 0.5 / (P.lazy(a_) + b_ + 3.0)
    w0, w1 = t.eval().min_max()
    for mode in (0, 2):
        with P.jit(mode):
            g0, g1 = t.min_max()
        assert (g0.bits(), g1.bits()) == (w0.bits(), w1.bits())
    # against the oracle on one case
    h0, h1 = host[eco.U16][:N], host[eco.U16][1:N + 1]
    vals = eco.f_binop(eco.DIV, eco.f_binop(eco.SUB, h0, h1), eco.f_binop(eco.ADD, h0, h1))
    emn, emx = eco.f_min_max(vals, None)
    with P.jit(2):
        mn, mx = P.program_min_max([dev[eco.U16].shard(0, N), dev[eco.U16].shard(1, N)], [], ndvi)
    assert (mn.bits(), mx.bits()) == (emn.bits(), emx.bits())


def test_compiled_programs_use_the_short_divide_only_where_it_is_exact(ec):
    """The generator turns a divide into the short exact divide when the program proves both operands integers of magnitude
    ≤ 131070 (≤ 16-bit integer cells, their sums and differences).  Compiled against interpreted (the IEEE expansion), compared
    on the device: every u8 / i8 pair, every u16 and i16 denominator against 2048 numerators (extremes included), NDVI-shaped
    sums and differences over the same operand sets — zero divisors and 0/0 included."""
    P, L = ec.fused, ec.lib()
    S, R = (lambda k: k), (lambda k: 4 + k)
    rng = np.random.default_rng(11)

    def grid_of(dt, rows):
        info = np.iinfo(dt)
        den = np.arange(info.min, info.max + 1, dtype=np.int64).astype(dt)
        num = np.concatenate([np.array([info.min, info.max, 0, 1, info.max - 1], dtype=np.int64),
                              rng.integers(info.min, info.max + 1, rows - 5)]).astype(dt)
        return np.repeat(num, den.size), np.tile(den, num.size)

    sets = []
    for dt in (np.uint8, np.int8):
        sets.append(grid_of(dt, 256 if dt == np.uint8 else 256))
    for dt in (np.uint16, np.int16):
        sets.append(grid_of(dt, 2048))
    a8, b8 = sets[0]
    sets.append((a8.astype(np.uint16) * 257, np.tile(np.arange(256, dtype=np.int8), 256)))  # mixed widths: u16 over i8
    divide = [(eco.DIV, S(0), S(1), 0)]
    ndvi = [(eco.SUB, S(0), S(1), 0), (eco.ADD, S(0), S(1), 1), (eco.DIV, R(0), R(1), 0)]
    for ha, hb in sets:
        da, db = ec.CellBuffer.from_vec(ha), ec.CellBuffer.from_vec(hb)
        for prog in (divide, ndvi):
            src = P.program_source([da.ct, db.ct], 0, prog)
            assert "= divs(" in src and " / " not in src[src.index("void run("):src.index("extern \"C\"")]
            with P.jit(2):
                compiled = P.program([da, db], [], prog)
            with P.jit(0):
                interpreted = P.program([da, db], [], prog)
            assert compiled == interpreted, (ha.dtype, hb.dtype, prog)
    # a wider cell type on either side, or a register that is not a sum / difference of cells: the IEEE divide stays
    assert "= divs(" not in P.program_source([ec.UInt16, ec.UInt32], 0, divide)
    assert "= divs(" not in P.program_source([ec.UInt16, ec.UInt16], 1, [(eco.MUL, S(0), 8, 0), (eco.DIV, R(0), S(1), 0)])
    assert "= divs(" not in P.program_source([ec.UInt16, ec.UInt16], 0, [(eco.SUB, S(0), S(1), 0), (eco.ADD, R(0), S(1), 1), (eco.DIV, R(1), S(0), 0)])


def test_expr_rejects_malformed_programs(ec, pool):
    host, dev, _, _ = pool
    L, E = ec.lib(), ec._ffi
    x = dev[eco.U16].shard(0, N)
    out = ec.CellBuffer.empty(N, ec.Float64)
    dt, p = (C.c_uint8 * 1)(eco.U16), (C.c_void_p * 1)(x.mem.ptr)
    sc = (E.EcValue * 1)(ec.CellValue.new(2.0).to_ec())

    def run(steps, n_streams=1, n_scalars=1):
        st = (E.EcExprStep * max(1, len(steps)))(*[E.EcExprStep(*s_) for s_ in steps])
        return L.ec_expr(dt, p, n_streams, sc, n_scalars, st, len(steps), N, out.mem.ptr, ec.stream())

    assert run([(eco.ADD, 0, 8, 0)]) == E.EC_OK
    assert run([(eco.ADD, 0, 4, 0)]) == E.EC_ERR_ARG          # register 0 read before any step wrote it
    assert run([(eco.ADD, 1, 8, 0)]) == E.EC_ERR_ARG          # stream 1 of a one-stream call
    assert run([(eco.ADD, 0, 9, 0)]) == E.EC_ERR_ARG          # scalar 1 of one
    assert run([(eco.ADD, 0, 8, 4)]) == E.EC_ERR_ARG          # register 4
    assert run([(7, 0, 8, 0)]) == E.EC_ERR_ARG                # op
    assert run([]) == E.EC_ERR_ARG
    assert run([(eco.ADD, 0, 8, 0)] * 17) == E.EC_ERR_ARG
    assert run([(eco.ADD, 0, 8, 0)], n_streams=5) == E.EC_ERR_ARG
    assert run([(eco.ADD, 0, 8, 0)], n_scalars=9) == E.EC_ERR_ARG
    bad_dt = (C.c_uint8 * 1)(10)
    st = (E.EcExprStep * 1)(E.EcExprStep(eco.ADD, 0, 8, 0))
    assert L.ec_expr(bad_dt, p, 1, sc, 1, st, 1, N, out.mem.ptr, ec.stream()) == E.EC_ERR_UNSUPPORTED_TYPE


SMALL = {  # cell type -> (lowest, highest) cell value
    "UInt8": (0, 255), "Int8": (-128, 127), "UInt16": (0, 65535), "Int16": (-32768, 32767),
}


class _OracleOnDevice:
    """Expected cells computed by the CPU oracle (all host cores) into a page-locked buffer, copied to the GPU and
    compared there with ec_buffer_cmp — so that billions of cells can be checked against the ORACLE (not against
    another HIP kernel) without downloading results."""

    def __init__(self, ec, cap):
        import os
        import torch
        self.ec, self.torch = ec, torch
        self.host = torch.empty(cap, dtype=torch.float64).pin_memory()
        self.dev = torch.empty(cap, dtype=torch.float64, device="cuda")
        self.cores = min(64, len(os.sched_getaffinity(0)))
        eco.set_threads(self.cores)

    def close(self):
        eco.set_threads(0)

    def expect_buffer(self, n):
        return self.host.numpy()[:n]

    def check(self, got_ptr, n, what):
        self.dev[:n].copy_(self.host[:n], non_blocking=False)
        order = C.c_int32(7)
        self.ec._ffi.check(self.ec.lib().ec_buffer_cmp(self.ec.Float64, got_ptr, n, self.ec.Float64, self.dev.data_ptr(), n,
                                                      C.byref(order), None))
        if order.value != 0:
            idx = C.c_uint64()
            self.ec._ffi.check(self.ec.lib().ec_first_difference(self.ec.Float64, got_ptr, self.dev.data_ptr(), n, C.byref(idx), None))
            raise AssertionError(f"{what}: first cell that differs from the oracle: {idx.value}")


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("lname", list(SMALL))
@pytest.mark.parametrize("rname", list(SMALL))
def test_small_integer_divide_is_exact_for_every_operand_pair(ec, lname, rname):
    """The divide of two cells of at most 16 bits uses a 6-instruction sequence (ec_device.hpp div_small_int) instead
    of the IEEE expansion.  Its operand space is small enough to try EVERY pair against the CPU ORACLE: for each of
    the 16 type pairs every (a, b) goes through ec_binop on the GPU, the oracle's typed loop (`(a as f64) / (b as
    f64)`, x86 divsd — src/value.rs:207) computes the same cells on the host, and the two are compared bit for bit on
    the device.  The four 16-bit pairs alone are 4 x 2^32 cells; together the pairs cover all 98304^2 operand values,
    zeros in the divisor (±inf, NaN with the x86 sign) included."""
    import torch
    L = ec.lib()
    chk = ec._ffi.check
    lt, rt = getattr(ec, lname), getattr(ec, rname)
    tdt = {"UInt8": torch.uint8, "Int8": torch.int8, "UInt16": torch.uint16, "Int16": torch.int16}
    (llo, lhi), (rlo, rhi) = SMALL[lname], SMALL[rname]
    na, nb = lhi - llo + 1, rhi - rlo + 1
    a_vals = torch.arange(llo, lhi + 1, dtype=torch.int32, device="cuda")
    ha_vals = np.arange(llo, lhi + 1).astype(ec.NP_DTYPES[lt])
    rows = max(1, min(nb, (1 << 27) // na))  # b-values per chunk: at most 2^27 cells at a time
    orc = _OracleOnDevice(ec, na * rows)
    try:
        for b0 in range(rlo, rhi + 1, rows):
            nb_c = min(rows, rhi + 1 - b0)
            n = na * nb_c
            a = a_vals.repeat(nb_c).to(tdt[lname])
            b = torch.arange(b0, b0 + nb_c, dtype=torch.int32, device="cuda").repeat_interleave(na).to(tdt[rname])
            fast = torch.empty(n, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            chk(L.ec_binop(ec.DIV, lt, a.data_ptr(), rt, b.data_ptr(), n, fast.data_ptr(), None))
            ha = np.tile(ha_vals, nb_c)
            hb = np.repeat(np.arange(b0, b0 + nb_c).astype(ec.NP_DTYPES[rt]), na)
            eco.f_binop(eco.DIV, ha, hb, orc.expect_buffer(n))
            orc.check(fast.data_ptr(), n, f"{lname} / {rname}, divisors from {b0}")
    finally:
        orc.close()
    # the reference-shaped form of the oracle agrees on a slice too (zeros in the divisor included)
    l = np.arange(llo, lhi + 1).astype(ec.NP_DTYPES[lt])[:4096]
    for bv in (0, 1, rlo, rhi, 3, 7):
        r = np.full(l.size, bv).astype(ec.NP_DTYPES[rt])
        got = (ec.CellBuffer.from_vec(l) / ec.CellBuffer.from_vec(r)).to_numpy()
        assert_f64_bits_equal(got, eco.binop(eco.DIV, l, r))


@pytest.mark.timeout(3000)
@pytest.mark.parametrize("tname", ["UInt16", "Int16"])
def test_fused_ndvi_small_integer_divide_is_exact_for_every_numerator_and_denominator(ec, tname):
    """`(x - y) / (z + w)` on 16-bit cells runs as 2 adds + the 6-instruction divide (ec_fused_kernels.hpp).  Every
    (numerator, denominator) such cells can produce is enumerated — 131071² pairs — and the fused kernel is compared
    with the ORACLE's eager chain (`f_binop` Sub, Add, then Div of the f64 temporaries: three x86 steps per cell,
    as src/gdal/rasterband.rs:148 evaluates it), bit for bit, on the device."""
    import torch
    L = ec.lib()
    chk = ec._ffi.check
    ct = getattr(ec, tname)
    tdt = {"UInt16": torch.uint16, "Int16": torch.int16}[tname]
    npdt = ec.NP_DTYPES[ct]
    lo, hi = SMALL[tname]
    t1_lo, t1_hi, t2_lo, t2_hi = lo - hi, hi - lo, 2 * lo, 2 * hi
    n1 = t1_hi - t1_lo + 1
    t1 = torch.arange(t1_lo, t1_hi + 1, dtype=torch.int32, device="cuda")
    if lo == 0:
        x1, y1 = t1.clamp(min=0), (-t1).clamp(min=0)
    else:
        x1 = torch.div(t1, 2, rounding_mode="floor")
        y1 = x1 - t1
    assert int(x1.min()) >= lo and int(x1.max()) <= hi and int(y1.min()) >= lo and int(y1.max()) <= hi
    hx1, hy1 = x1.cpu().numpy().astype(npdt), y1.cpu().numpy().astype(npdt)
    rows = max(1, (1 << 27) // n1)
    dt4 = (C.c_uint8 * 4)(ct, ct, ct, ct)
    orc = _OracleOnDevice(ec, n1 * rows)
    e1, e2 = np.empty(n1 * rows, np.float64), np.empty(n1 * rows, np.float64)
    try:
        for b0 in range(t2_lo, t2_hi + 1, rows):
            nb = min(rows, t2_hi + 1 - b0)
            n = n1 * nb
            t2 = torch.arange(b0, b0 + nb, dtype=torch.int32, device="cuda")
            z1 = torch.div(t2, 2, rounding_mode="floor")
            w1 = t2 - z1
            x, y = x1.repeat(nb).to(tdt), y1.repeat(nb).to(tdt)
            z, w = z1.repeat_interleave(n1).to(tdt), w1.repeat_interleave(n1).to(tdt)
            fused = torch.empty(n, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            p4 = (C.c_void_p * 4)(x.data_ptr(), y.data_ptr(), z.data_ptr(), w.data_ptr())
            chk(L.ec_fused(ec.SUB, ec.DIV, ec.ADD, dt4, p4, None, n, fused.data_ptr(), None))
            hz1 = z1.cpu().numpy().astype(npdt)
            hw1 = w1.cpu().numpy().astype(npdt)
            eco.f_binop(eco.SUB, np.tile(hx1, nb), np.tile(hy1, nb), e1[:n])
            eco.f_binop(eco.ADD, np.repeat(hz1, n1), np.repeat(hw1, n1), e2[:n])
            eco.f_binop(eco.DIV, e1[:n], e2[:n], orc.expect_buffer(n))
            orc.check(fused.data_ptr(), n, f"fused NDVI {tname}, denominators from {b0}")
    finally:
        orc.close()


# ---- the ahead-of-time catalogue of expression programs (csrc/ec_expr_fixed.hpp): 4 formulas x 4 cell widths = 16 kernels
_S, _R, _K = (lambda k: k), (lambda k: 4 + k), (lambda k: 8 + k)
_FIXED = {
    "ndvi": (2, [], [[(eco.SUB, _S(0), _S(1), 0), (eco.ADD, _S(0), _S(1), 1), (eco.DIV, _R(0), _R(1), 0)],
                     [(eco.ADD, _S(0), _S(1), 3), (eco.SUB, _S(0), _S(1), 2), (eco.DIV, _R(2), _R(3), 1)]]),
    "add-mul": (3, [], [[(eco.ADD, _S(0), _S(1), 0), (eco.MUL, _R(0), _S(2), 0)],
                        [(eco.ADD, _S(0), _S(1), 2), (eco.MUL, _R(2), _S(2), 3)]]),
    "evi": (3, [2.5, 6.0, 7.5, 1.0],
            [[(eco.SUB, _S(0), _S(1), 0), (eco.MUL, _R(0), _K(0), 0), (eco.MUL, _S(1), _K(1), 1), (eco.ADD, _S(0), _R(1), 1),
              (eco.MUL, _S(2), _K(2), 2), (eco.SUB, _R(1), _R(2), 1), (eco.ADD, _R(1), _K(3), 1), (eco.DIV, _R(0), _R(1), 0)],
             # denominator first, scalars on the left of their products
             [(eco.MUL, _K(1), _S(1), 3), (eco.ADD, _S(0), _R(3), 3), (eco.MUL, _K(2), _S(2), 0), (eco.SUB, _R(3), _R(0), 2),
              (eco.ADD, _R(2), _K(3), 2), (eco.SUB, _S(0), _S(1), 1), (eco.MUL, _K(0), _R(1), 1), (eco.DIV, _R(1), _R(2), 0)]]),
    "affine": (1, [0.0001, -273.15], [[(eco.MUL, _S(0), _K(0), 0), (eco.ADD, _R(0), _K(1), 0)],
                                      [(eco.MUL, _K(0), _S(0), 2), (eco.ADD, _K(1), _R(2), 1)]]),
}


@pytest.mark.gpu
def test_expr_ahead_of_time_kernels_every_formula_every_width(ec, pool):
    """All 16 `k_expr_fixed<formula, width>` kernels, every cell kind inside each width: a catalogue formula — in two spellings
    each (other registers, other schedule, scalars on the other side) — runs as ONE launch of its built-in kernel
    (`expr_fixed_launches`), with expr_jit = 0 and no interpreter launch, and gives the oracle's step-by-step cells and, bit for
    bit, the interpreter's (`expr_fixed` = 0) — over whole tiles, a ragged tail, odd windows (peeled head, odd last cell), one
    and two cells, plain and masked."""
    host, dev, m, dm = pool
    P, L = ec.fused, ec.lib()
    L.ec_tune_set(b"expr_jit", 0)
    try:
        tick = 0
        for name, (ns, scalars, spellings) in _FIXED.items():
            for width, cts_of in _BY_WIDTH.items():
                for rot in range(len(cts_of)):
                    cts = [cts_of[(rot + k) % len(cts_of)] for k in range(ns)]  # kinds differ between the streams of one launch
                    for n, off in ((N, 0), (2049, 3), (1, 0), (2, 1)):
                        tick += 1
                        bufs = [dev[ct].shard(off + k, n) for k, ct in enumerate(cts)]
                        hs = [host[ct][off + k:off + k + n] for k, ct in enumerate(cts)]
                        steps = spellings[tick % 2]
                        f0, i0 = _stat(ec, b"expr_fixed_launches"), _stat(ec, b"expr_interp_launches")
                        masked = tick % 3 == 0
                        if masked:
                            got_m = P.program([ec.MaskedCellBuffer(b, dm[k % 2].shard(off + k, n)) for k, b in enumerate(bufs)], scalars, steps)
                            got = got_m.buffer()
                            want_mask = np.ones(n, np.uint8)
                            for k in range(ns):
                                want_mask &= m[k % 2][off + k:off + k + n]
                            assert np.array_equal(got_m.mask().to_numpy(), want_mask)
                        else:
                            got = P.program(bufs, scalars, steps)
                        assert _stat(ec, b"expr_fixed_launches") == f0 + 1 and _stat(ec, b"expr_interp_launches") == i0, (name, cts, n)
                        eo, loose = _oracle_program(hs, scalars, spellings[0])
                        try:
                            assert_f64_bits_equal(got.to_numpy(), eo, nan_by_class_where=loose)
                        except AssertionError as e:
                            raise AssertionError(f"{name} types {cts} n {n} off {off}: {e}") from None
                        L.ec_tune_set(b"expr_fixed", 0)
                        try:
                            interp = P.program(bufs, scalars, steps)
                            assert _stat(ec, b"expr_interp_launches") == i0 + 1
                        finally:
                            L.ec_tune_set(b"expr_fixed", 1)
                        assert np.array_equal(bits_of(got.to_numpy()), bits_of(interp.to_numpy())), (name, cts, n, off)
    finally:
        L.ec_tune_set(b"expr_jit", 1)


@pytest.mark.gpu
def test_expr_ahead_of_time_near_misses_stay_with_the_interpreter(ec, pool):
    """What is NOT served by a built-in kernel, and must still be right: a catalogue tree over streams of different widths, one
    buffer under two stream names, a NaN scalar on the left of its product (the swap that makes the tree canonical would change
    which NaN wins), operands of a subtraction swapped — all interpreted (expr_jit = 0), all equal to the oracle; and special
    scalars ON the catalogue's path (inf, -0.0, NaN on the right) against the interpreter bit for bit."""
    host, dev, m, dm = pool
    P, L = ec.fused, ec.lib()
    L.ec_tune_set(b"expr_jit", 0)
    try:
        ndvi = _FIXED["ndvi"][2][0]
        affine_left = _FIXED["affine"][2][1]  # k0 * s0, k1 + r
        cases = [([eco.U16, eco.F32], [], ndvi),                                        # widths differ
                 ([eco.U16, eco.U16], [], [(eco.SUB, _S(1), _S(0), 0), (eco.ADD, _S(0), _S(1), 1), (eco.DIV, _R(0), _R(1), 0)]),  # another tree
                 ([eco.F32], [float("nan"), 1.0], affine_left)]                          # NaN scalar on the left: not swapped
        for cts, scalars, steps in cases:
            bufs = [dev[ct].shard(k, N) for k, ct in enumerate(cts)]
            hs = [host[ct][k:k + N] for k, ct in enumerate(cts)]
            f0, i0 = _stat(ec, b"expr_fixed_launches"), _stat(ec, b"expr_interp_launches")
            got = P.program(bufs, scalars, steps)
            assert _stat(ec, b"expr_fixed_launches") == f0 and _stat(ec, b"expr_interp_launches") == i0 + 1, (cts, steps)
            eo, loose = _oracle_program(hs, scalars, steps)
            assert_f64_bits_equal(got.to_numpy(), eo, nan_by_class_where=loose)
        # one buffer as both streams of NDVI: (a - a) / (a + a)
        a = dev[eco.I16].shard(0, N)
        f0 = _stat(ec, b"expr_fixed_launches")
        got = P.program([a, a], [], ndvi)
        assert _stat(ec, b"expr_fixed_launches") == f0
        eo, loose = _oracle_program([host[eco.I16][:N], host[eco.I16][:N]], [], ndvi)
        assert_f64_bits_equal(got.to_numpy(), eo, nan_by_class_where=loose)
        # special scalars through the built-in kernels
        for scalars in ([float("inf"), 1.0], [-0.0, float("-inf")], [2.0, float("nan")], [0.0, 0.0]):
            for ct in (eco.F32, eco.F64, eco.U8, eco.I64):
                x = dev[ct].shard(1, N)
                f0 = _stat(ec, b"expr_fixed_launches")
                got = P.program([x], scalars, _FIXED["affine"][2][0])
                assert _stat(ec, b"expr_fixed_launches") == f0 + 1
                L.ec_tune_set(b"expr_fixed", 0)
                try:
                    interp = P.program([x], scalars, _FIXED["affine"][2][0])
                finally:
                    L.ec_tune_set(b"expr_fixed", 1)
                assert np.array_equal(bits_of(got.to_numpy()), bits_of(interp.to_numpy())), (scalars, ct)
                eo, loose = _oracle_program([host[ct][1:1 + N]], scalars, _FIXED["affine"][2][0])
                assert_f64_bits_equal(got.to_numpy(), eo, nan_by_class_where=loose)
    finally:
        L.ec_tune_set(b"expr_jit", 1)


@pytest.mark.gpu
def test_expr_ahead_of_time_kernel_inside_a_stream_capture_and_from_lazy_trees(ec, pool):
    """A catalogue program first met INSIDE a stream capture is recorded as its built-in kernel (the compiled form cannot be:
    no module load in a capture), and the operator syntax reaches the same kernels: `2.5 * (n - r) / (n + 6.0 * r - 7.5 * b + 1.0)`
    written with `lazy()` is one launch of the EVI kernel."""
    import torch
    host, dev, m, dm = pool
    P, L, E = ec.fused, ec.lib(), ec._ffi
    nir, red, blue = dev[eco.U16].shard(0, N), dev[eco.I16].shard(2, N), dev[eco.U16].shard(4, N)
    hn, hr, hb = host[eco.U16][:N], host[eco.I16][2:2 + N], host[eco.U16][4:4 + N]
    from erased_cells_hip.fused import lazy
    f0 = _stat(ec, b"expr_fixed_launches")
    got = (2.5 * (lazy(nir) - red) / (lazy(nir) + 6.0 * lazy(red) - 7.5 * lazy(blue) + 1.0)).eval()
    assert _stat(ec, b"expr_fixed_launches") == f0 + 1
    eo, loose = _oracle_program([hn, hr, hb], _FIXED["evi"][1], _FIXED["evi"][2][0])
    assert_f64_bits_equal(got.to_numpy(), eo, nan_by_class_where=loose)
    # inside a capture
    out = ec.CellBuffer.empty(N, ec.Float64)
    steps = _FIXED["evi"][2][1]
    dt = (C.c_uint8 * 3)(eco.U16, eco.I16, eco.U16)
    p = (C.c_void_p * 3)(nir.mem.ptr, red.mem.ptr, blue.mem.ptr)
    sc = (E.EcValue * 4)(*[ec.CellValue.new(x).to_ec() for x in _FIXED["evi"][1]])
    st = (E.EcExprStep * len(steps))(*[E.EcExprStep(op, a, b, dst) for op, a, b, dst in steps])
    cap = torch.cuda.Stream()
    E.check(L.ec_prepare_stream(cap.cuda_stream))
    g = torch.cuda.CUDAGraph()
    cap.wait_stream(torch.cuda.current_stream())
    f0, i0 = _stat(ec, b"expr_fixed_launches"), _stat(ec, b"expr_interp_launches")
    with torch.cuda.graph(g, stream=cap):
        E.check(L.ec_expr(dt, p, 3, sc, 4, st, len(steps), N, out.mem.ptr, cap.cuda_stream))
    assert _stat(ec, b"expr_fixed_launches") == f0 + 1 and _stat(ec, b"expr_interp_launches") == i0
    g.replay()
    torch.cuda.synchronize()
    assert_f64_bits_equal(out.to_numpy(), eo, nan_by_class_where=loose)
