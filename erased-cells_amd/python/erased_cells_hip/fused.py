"""Fused operator chains: one pass over HBM for `(x o1 y) o2 (z o3 w)`.

The reference evaluates `(&nir - &red) / (nir + red)` (src/gdal/rasterband.rs:148) or
`(buf + ones) * 2.0`-style chains eagerly, writing an f64 temporary per operator.  These helpers
produce bit-identical results (each step is the same rounded f64 op) from a single kernel.
"""
from __future__ import annotations

import ctypes as C

from . import buffer as B
from ._ffi import check, lib

OP_NONE = -1


def _is_buf(o) -> bool:
    return isinstance(o, (B.CellBuffer, B.MaskedCellBuffer))


def expr(x, o1: int, y, o2: int, z, o3: int = OP_NONE, w=None):
    """(x o1 y) o2 (z o3 w)  — or (x o1 y) o2 z when o3 is OP_NONE.  Operands are all CellBuffer or all
    MaskedCellBuffer; any of them may instead be a scalar (number / CellValue), e.g. `(buf + ones) * 2.0`."""
    ops = [x, y, z] + ([w] if o3 != OP_NONE else [])
    bufs_in = [o for o in ops if _is_buf(o)]
    assert bufs_in, "at least one operand must be a buffer"
    masked = isinstance(bufs_in[0], B.MaskedCellBuffer)
    assert all(isinstance(o, B.MaskedCellBuffer) == masked for o in bufs_in), "mix of masked and plain operands"
    dt = (C.c_uint8 * 4)()
    p = (C.c_void_p * 4)()
    m = (C.c_void_p * 4)()
    sc = (B.EcValue * 4)()
    for k, o in enumerate(ops):
        if _is_buf(o):
            b = o.buffer() if masked else o
            dt[k], p[k] = b.ct, b.mem.ptr
            if masked:
                m[k] = o.mask().mem.ptr
        else:
            sc[k] = B.CellValue.new(o).to_ec()
    n = min((o.buffer() if masked else o).len() for o in bufs_in)  # zip truncation of every step (src/buffer.rs:327)
    if n == 0:
        e = B.CellBuffer.empty(0, B.UInt8)
        return B.MaskedCellBuffer(e, B.Mask.empty(0)) if masked else e
    out = B.CellBuffer.empty(n, B.Float64)
    if not masked:
        check(lib().ec_fused(o1, o2, o3, dt, p, sc, n, out.mem.ptr, B.stream()))
        return out
    om = B.Mask.empty(n)
    check(lib().ec_masked_fused(o1, o2, o3, dt, p, m, sc, n, out.mem.ptr, om.mem.ptr, B.stream()))
    return B.MaskedCellBuffer(out, om)


def ndvi(nir, red):
    """(nir - red) / (nir + red), one pass (src/gdal/rasterband.rs:148,178)."""
    return expr(nir, B.SUB, red, B.DIV, nir, B.ADD, red)


def add_mul(a, b, c):
    """(a + b) * c, one pass (BASELINE config 3)."""
    return expr(a, B.ADD, b, B.MUL, c)
