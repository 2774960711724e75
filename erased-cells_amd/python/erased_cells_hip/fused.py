"""Fused operator chains: one pass over HBM for `(x o1 y) o2 (z o3 w)` — and for operator trees of any depth.

The reference evaluates `(&nir - &red) / (nir + red)` (src/gdal/rasterband.rs:148) or
`(buf + ones) * 2.0`-style chains eagerly, writing an f64 temporary per operator.  These helpers
produce bit-identical results (each step is the same rounded f64 op) from a single kernel.

  expr / ndvi / chain3          the two-level shapes (`ec_fused`)
  program(streams, scalars, steps)   an expression program: ≤ 4 buffers, 8 scalars, 16 steps over 4 registers (`ec_expr`)
  lazy(buf)                      operator syntax; trees are scheduled onto a program (or the two-level kernel) by `eval()`
  jit(mode)                      how programs run: interpreted / compiled in the background (default) / compiled at once
  program_host, program_host_masked, pinned_empty    numpy arrays in, numpy array out, streamed over PCIe (`ec_host_expr`)
  program_min_max                (min, max) of a program's result without its raster (`ec_expr_min_max`)
  program_source                 the HIP source the library compiles for a program (no GPU needed)
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


STREAM0, REG0, SCALAR0 = 0, 4, 8  # operand references of a program step (EC_EXPR_STREAM / _REG / _SCALAR)
MAX_STREAMS, REGS, MAX_SCALARS, MAX_STEPS = 4, 4, 8, 16


def program(streams, scalars, steps):
    """Run an expression PROGRAM in one pass (`ec_expr` / `ec_masked_expr`): `streams` up to four buffers (all plain or
    all masked), `scalars` up to eight numbers, `steps` a list of `(op, a, b, dst)` with `a`, `b` references
    (`STREAM0 + k`, `REG0 + k`, `SCALAR0 + k`) and `dst` a register 0..3; the value of the program is what its last step
    computed.  Bit-identical to evaluating the same operators one by one."""
    assert 1 <= len(streams) <= MAX_STREAMS and len(scalars) <= MAX_SCALARS and 1 <= len(steps) <= MAX_STEPS
    masked = isinstance(streams[0], B.MaskedCellBuffer)
    assert all(isinstance(o, B.MaskedCellBuffer) == masked for o in streams), "mix of masked and plain operands"
    bufs = [o.buffer() if masked else o for o in streams]
    k = len(streams)
    dt = (C.c_uint8 * k)(*[b.ct for b in bufs])
    p = (C.c_void_p * k)(*[b.mem.ptr for b in bufs])
    sc = (B.EcValue * max(1, len(scalars)))(*[B.CellValue.new(x).to_ec() for x in scalars])
    from ._ffi import EcExprStep
    st = (EcExprStep * len(steps))(*[EcExprStep(*s_) for s_ in steps])
    n = min(b.len() for b in bufs)
    if n == 0:
        e = B.CellBuffer.empty(0, B.UInt8)
        return B.MaskedCellBuffer(e, B.Mask.empty(0)) if masked else e
    out = B.CellBuffer.empty(n, B.Float64)
    if not masked:
        check(lib().ec_expr(dt, p, k, sc, len(scalars), st, len(steps), n, out.mem.ptr, B.stream()))
        return out
    m = (C.c_void_p * k)(*[o.mask().mem.ptr for o in streams])
    om = B.Mask.empty(n)
    check(lib().ec_masked_expr(dt, p, m, k, sc, len(scalars), st, len(steps), n, out.mem.ptr, om.mem.ptr, B.stream()))
    return B.MaskedCellBuffer(out, om)


class jit:
    """`with fused.jit(2): ...` — how expression programs are run inside the block (`ec_tune_set("expr_jit", mode)`): 0 the
    interpreter kernel only, 1 compiled in the background once a program has run long enough (the default), 2 compiled on the
    calling thread at first sight.  The mode that was in force before the block is restored on exit (blocks nest)."""

    def __init__(self, mode: int):
        self.mode = mode
        self.previous = 1

    def __enter__(self):
        prev = C.c_int64(1)
        check(lib().ec_stat_get(b"tune.expr_jit", C.byref(prev)))
        self.previous = prev.value
        check(lib().ec_tune_set(b"expr_jit", self.mode))
        return self

    def __exit__(self, *exc):
        check(lib().ec_tune_set(b"expr_jit", self.previous))
        return False


def pinned_empty(n: int, dtype):
    """A page-locked numpy array (`ec_host_alloc`): what `program_host` copies to and from without registering pages first.
    The memory is returned to the system when the array (and every view of it) is garbage."""
    import numpy as np
    import weakref
    dt = np.dtype(dtype)
    p = C.c_void_p()
    check(lib().ec_host_alloc(C.byref(p), max(1, n * dt.itemsize)))
    buf = (C.c_char * max(1, n * dt.itemsize)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dt, count=n)
    weakref.finalize(buf, lib().ec_host_free, p)
    return arr


def program_host(arrays, scalars, steps, out=None, chunk_cells: int = 0):
    """An expression program over HOST arrays (numpy, one per stream, any of the ten cell types), streamed through the GPU
    in chunks with upload, kernel and download overlapped (`ec_host_expr`): what the reference's Vec-in / Vec-out operators
    cost when nothing stays resident — PCIe-bound.  Returns (or fills `out`: a C-contiguous float64 array, ideally from
    `pinned_empty`) the f64 result of min(len) cells."""
    import numpy as np
    from ._ffi import EcExprStep
    arrays = [np.ascontiguousarray(a) for a in arrays]
    n = min(a.size for a in arrays)
    k = len(arrays)
    dt = (C.c_uint8 * k)(*[B.cell_type_of(a.dtype) for a in arrays])
    p = (C.c_void_p * k)(*[a.ctypes.data for a in arrays])
    sc = (B.EcValue * max(1, len(scalars)))(*[B.CellValue.new(x).to_ec() for x in scalars])
    st = (EcExprStep * len(steps))(*[EcExprStep(*s_) for s_ in steps])
    if out is None:
        out = np.empty(n, dtype=np.float64)
    assert out.dtype == np.float64 and out.flags.c_contiguous and out.size >= n
    check(lib().ec_host_expr(dt, p, k, sc, len(scalars), st, len(steps), n, out.ctypes.data, chunk_cells))
    return out[:n]


def program_host_masked(arrays, nodata, scalars, steps, out_nodata=None, out=None, want_mask=False, chunk_cells: int = 0):
    """The masked form of `program_host` (`ec_host_masked_expr`): `nodata[k]` is stream k's nodata value (None: the stream has
    none) — the masks are derived, ANDed and applied on the device while the chunks stream through.  Returns the f64 result with
    `out_nodata` in the cells that are not valid (None: the values of all cells), and the result's mask too with `want_mask`."""
    import numpy as np
    from ._ffi import EcExprStep, EcValue
    arrays = [np.ascontiguousarray(a) for a in arrays]
    n = min(a.size for a in arrays)
    k = len(arrays)
    cts = [B.cell_type_of(a.dtype) for a in arrays]
    dt = (C.c_uint8 * k)(*cts)
    p = (C.c_void_p * k)(*[a.ctypes.data for a in arrays])
    nds = [None if v is None else B.CellValue(ct, v).to_ec() for ct, v in zip(cts, nodata)]
    nd = (C.POINTER(EcValue) * k)(*[C.pointer(v) if v is not None else C.POINTER(EcValue)() for v in nds])
    sc = (EcValue * max(1, len(scalars)))(*[B.CellValue.new(x).to_ec() for x in scalars])
    st = (EcExprStep * len(steps))(*[EcExprStep(*s_) for s_ in steps])
    if out is None:
        out = np.empty(n, dtype=np.float64)
    assert out.dtype == np.float64 and out.flags.c_contiguous and out.size >= n
    mask = np.empty(n, dtype=np.uint8) if want_mask else None
    ond = C.c_double(out_nodata) if out_nodata is not None else None
    check(lib().ec_host_masked_expr(dt, p, nd, k, sc, len(scalars), st, len(steps), n, out.ctypes.data, C.byref(ond) if ond is not None else None,
                                    mask.ctypes.data if mask is not None else None, chunk_cells))
    return (out[:n], mask.astype(bool)) if want_mask else out[:n]


def program_source(cell_types, n_scalars, steps, arch=None) -> str:
    """The HIP source the library compiles for a program when it compiles it for itself (`ec_expr_source`; no GPU needed).
    With `arch` (e.g. "gfx950") the source is also compiled once with hiprtc; a failure raises with the compiler's log."""
    from ._ffi import EcExprStep
    dt = (C.c_uint8 * len(cell_types))(*cell_types)
    st = (EcExprStep * len(steps))(*[EcExprStep(*s_) for s_ in steps])
    need = C.c_size_t(0)
    check(lib().ec_expr_source(dt, len(cell_types), n_scalars, st, len(steps), None, None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    check(lib().ec_expr_source(dt, len(cell_types), n_scalars, st, len(steps), arch.encode() if arch else None, buf, need.value, C.byref(need)))
    return buf.value.decode()


def program_min_max(streams, scalars, steps):
    """`(min, max)` of a program's result without its raster (`ec_expr_min_max`): over all cells of plain buffers, over the
    valid cells (AND of the masks) of masked ones.  Once the library has compiled the program for itself only the streams are
    read; until then it runs the program into a temporary and reduces that."""
    from ._ffi import EcExprStep
    masked = isinstance(streams[0], B.MaskedCellBuffer)
    assert all(isinstance(o, B.MaskedCellBuffer) == masked for o in streams), "mix of masked and plain operands"
    bufs = [o.buffer() if masked else o for o in streams]
    k = len(streams)
    dt = (C.c_uint8 * k)(*[b.ct for b in bufs])
    p = (C.c_void_p * k)(*[b.mem.ptr for b in bufs])
    m = (C.c_void_p * k)(*[o.mask().mem.ptr for o in streams]) if masked else None
    sc = (B.EcValue * max(1, len(scalars)))(*[B.CellValue.new(x).to_ec() for x in scalars])
    st = (EcExprStep * len(steps))(*[EcExprStep(*s_) for s_ in steps])
    n = min(b.len() for b in bufs)
    mn, mx = B.EcValue(), B.EcValue()
    check(lib().ec_expr_min_max(dt, p, m, k, sc, len(scalars), st, len(steps), n, C.byref(mn), C.byref(mx), B.stream()))
    return B.CellValue.from_ec(mn), B.CellValue.from_ec(mx)


class _Compiler:
    """Schedules an operator tree onto the four registers of `ec_expr`: post-order, the sub-tree that needs more
    registers first (Sethi-Ullman), a register freed as soon as its value has been consumed.  `compile` returns None when
    the tree does not fit (more than 4 distinct buffers, 8 scalars, 16 operators or 4 live temporaries)."""

    def __init__(self):
        self.streams, self.scalars, self.steps, self.free = [], [], [], [0, 1, 2, 3]

    def _leaf(self, x):
        if _is_buf(x):
            b = x.buffer() if isinstance(x, B.MaskedCellBuffer) else x
            for i, s_ in enumerate(self.streams):
                t = s_.buffer() if isinstance(s_, B.MaskedCellBuffer) else s_
                same_mask = (not isinstance(x, B.MaskedCellBuffer)) or s_.mask().mem.ptr == x.mask().mem.ptr
                if t.mem.ptr == b.mem.ptr and t.ct == b.ct and t.len() == b.len() and same_mask:
                    return STREAM0 + i
            if len(self.streams) == MAX_STREAMS:
                raise OverflowError
            self.streams.append(x)
            return STREAM0 + len(self.streams) - 1
        if len(self.scalars) == MAX_SCALARS:
            raise OverflowError
        self.scalars.append(x)
        return SCALAR0 + len(self.scalars) - 1

    @staticmethod
    def _need(t) -> int:  # registers needed to evaluate the sub-tree
        if t.op is None:
            return 0
        l, r = _Compiler._need(t.l), _Compiler._need(t.r)
        return max(1, l, r) if l != r else l + 1 if l else 1

    def _emit(self, t):
        if t.op is None:
            return self._leaf(t.leaf)
        first_right = self._need(t.r) > self._need(t.l)
        if first_right:
            rb = self._emit(t.r)
            ra = self._emit(t.l)
        else:
            ra = self._emit(t.l)
            rb = self._emit(t.r)
        for ref in (ra, rb):  # operands that are registers are dead after this step
            if REG0 <= ref < SCALAR0 and ref - REG0 not in self.free:
                self.free.append(ref - REG0)
        if not self.free or len(self.steps) == MAX_STEPS:
            raise OverflowError
        dst = min(self.free)
        self.free.remove(dst)
        self.steps.append((t.op, ra, rb, dst))
        return REG0 + dst

    def compile(self, tree):
        try:
            self._emit(tree)
        except OverflowError:
            return None
        return self.streams, self.scalars, self.steps


class Lazy:
    """Operator syntax that defers evaluation, so a chain written the reference's way runs fused:

        n, r = lazy(nir), lazy(red)
        ndvi = ((n - r) / (n + r)).eval()        # one pass, bit-identical to (nir - red) / (nir + red)

    A tree of up to two levels — `(x o1 y) o2 z`, `(x o1 y) o2 (z o3 w)` — runs as one launch of the two-level kernel.
    A deeper tree runs as ONE launch of the expression-program kernel (`ec_expr`) when it has at most 4 distinct
    buffers, 8 scalars, 16 operators and can be scheduled onto 4 temporaries (EVI: 3 bands, 4 scalars, 8 operators, 2
    temporaries); only trees beyond that are cut: their deeper sub-trees are evaluated first, each again fused as far as
    it goes.  Leaves are buffers (all plain or all masked) or scalars.
    The C++ mirror has the same as expression templates (`lazy()`, host/erased_cells.hpp)."""

    __slots__ = ("op", "l", "r", "leaf")

    def __init__(self, leaf=None, op=None, l=None, r=None):
        self.leaf, self.op, self.l, self.r = leaf, op, l, r

    @staticmethod
    def _wrap(x) -> "Lazy":
        return x if isinstance(x, Lazy) else Lazy(leaf=x)

    def _node(self, op, other, swap=False):
        a, b = (Lazy._wrap(other), self) if swap else (self, Lazy._wrap(other))
        return Lazy(op=op, l=a, r=b)

    def __add__(self, o): return self._node(B.ADD, o)
    def __sub__(self, o): return self._node(B.SUB, o)
    def __mul__(self, o): return self._node(B.MUL, o)
    def __truediv__(self, o): return self._node(B.DIV, o)
    def __radd__(self, o): return self._node(B.ADD, o, swap=True)
    def __rsub__(self, o): return self._node(B.SUB, o, swap=True)
    def __rmul__(self, o): return self._node(B.MUL, o, swap=True)
    def __rtruediv__(self, o): return self._node(B.DIV, o, swap=True)

    def _depth(self) -> int:
        return 0 if self.op is None else 1 + max(self.l._depth(), self.r._depth())

    def _flat(self) -> "Lazy":
        """This sub-tree as a leaf (evaluating it if it is not one already)."""
        return self if self.op is None else Lazy(leaf=self.eval())

    def _has_buffer(self) -> bool:
        return _is_buf(self.leaf) if self.op is None else (self.l._has_buffer() or self.r._has_buffer())

    def min_max(self):
        """`(min, max)` of the tree's result without its raster when the tree fits one expression program (`program_min_max`);
        otherwise `eval().min_max()`."""
        if self.op is not None and self._has_buffer():
            prog = _Compiler().compile(self)
            if prog is not None and prog[0]:
                return program_min_max(*prog)
        return self.eval().min_max()

    def eval(self):
        if self.op is None:
            return self.leaf
        two_level = self._depth() == 2 and self.l.op is not None   # (x o1 y) o2 z  or  (x o1 y) o2 (z o3 w)
        if self._depth() >= 2 and not two_level and self._has_buffer():
            # not a shape of the two-level kernel (deeper, or `z o2 (x o1 y)`): one pass through the expression-program
            # kernel if the tree can be scheduled onto its four registers; otherwise the pieces below, deeper sub-trees first
            prog = _Compiler().compile(self)
            if prog is not None and prog[0]:
                return program(*prog)
        l = self.l if self.l._depth() <= 1 else self.l._flat()
        r = self.r if self.r._depth() <= 1 else self.r._flat()
        if l.op is None and r.op is None:           # x o y
            a, b = l.leaf, r.leaf
            if _is_buf(a):
                return a._binop(self.op, b)
            if _is_buf(b):                            # scalar o buffer: only the buffer-on-the-left form exists eagerly
                return expr(a, self.op, b, B.MUL, 1.0)  # (s o b) * 1.0 — exact, the product by one is the identity in f64
            return B.CellValue.new(a)._scalar_op(b, {B.ADD: lambda p, q: p + q, B.SUB: lambda p, q: p - q,
                                                      B.MUL: lambda p, q: p * q, B.DIV: lambda p, q: p / q}[self.op])
        if l.op is not None and r.op is None:        # (x o1 y) o2 z
            if _is_buf(l.l.leaf) or _is_buf(l.r.leaf) or _is_buf(r.leaf):
                return expr(l.l.leaf, l.op, l.r.leaf, self.op, r.leaf)
        if l.op is not None and r.op is not None:    # (x o1 y) o2 (z o3 w)
            leaves = (l.l.leaf, l.r.leaf, r.l.leaf, r.r.leaf)
            if any(_is_buf(x) for x in leaves):
                return expr(leaves[0], l.op, leaves[1], self.op, leaves[2], r.op, leaves[3])
        # z o2 (x o1 y), or sub-trees without buffers: evaluate the children, then one plain op
        return Lazy(op=self.op, l=l._flat(), r=r._flat()).eval()


def lazy(x) -> Lazy:
    return Lazy(leaf=x)
