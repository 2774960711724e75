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


def _arrays(ops):
    n = len(ops)
    dt = (C.c_uint8 * 4)(*([o.cell_type() for o in ops] + [0] * (4 - n)))
    bufs = [o.buffer() if isinstance(o, B.MaskedCellBuffer) else o for o in ops]
    p = (C.c_void_p * 4)(*([b.mem.ptr for b in bufs] + [None] * (4 - n)))
    return dt, p, bufs


def expr(x, o1: int, y, o2: int, z, o3: int = OP_NONE, w=None):
    """(x o1 y) o2 (z o3 w)  — or (x o1 y) o2 z when o3 is OP_NONE.  All CellBuffer, or all MaskedCellBuffer."""
    ops = [x, y, z] + ([w] if o3 != OP_NONE else [])
    masked = isinstance(x, B.MaskedCellBuffer)
    assert all(isinstance(o, B.MaskedCellBuffer) == masked for o in ops), "mix of masked and plain operands"
    dt, p, bufs = _arrays(ops)
    n = min(b.len() for b in bufs)  # zip truncation of every step (src/buffer.rs:327)
    if n == 0:
        e = B.CellBuffer.empty(0, B.UInt8)
        return B.MaskedCellBuffer(e, B.Mask.empty(0)) if masked else e
    out = B.CellBuffer.empty(n, B.Float64)
    if not masked:
        check(lib().ec_fused(o1, o2, o3, dt, p, n, out.mem.ptr, B.stream()))
        return out
    m = (C.c_void_p * 4)(*([o.mask().mem.ptr for o in ops] + [None] * (4 - len(ops))))
    om = B.Mask.empty(n)
    check(lib().ec_masked_fused(o1, o2, o3, dt, p, m, n, out.mem.ptr, om.mem.ptr, B.stream()))
    return B.MaskedCellBuffer(out, om)


def ndvi(nir, red):
    """(nir - red) / (nir + red), one pass (src/gdal/rasterband.rs:148,178)."""
    return expr(nir, B.SUB, red, B.DIV, nir, B.ADD, red)


def add_mul(a, b, c):
    """(a + b) * c, one pass (BASELINE config 3)."""
    return expr(a, B.ADD, b, B.MUL, c)
