"""Python host mirror of the reference's buffer API over the C ABI.

Same names, argument meaning and error behaviour as `CellBuffer` /
`MaskedCellBuffer` / `Mask` / `NoData` / `CellValue` / `CellType`
(src/buffer.rs, src/masked/*.rs, src/value.rs, src/ctype.rs), so that the parity
tests read like the reference's own tests.  Buffers live in HBM between
operations (`from_vec` uploads once, `to_vec` downloads); every operator body is
one call into liberased_cells_hip.so.  The host keeps what the reference's host
code keeps: the dtype-erased dispatch tag, zip truncation (src/buffer.rs:327),
the empty-result-is-UInt8 rule (src/buffer.rs:233-234) and the length asserts.

This module is test/bench plumbing; the compiled host mirror is
erased-cells_amd/host/erased_cells.hpp and the Rust binding is in INTEGRATION.md.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Sequence

import numpy as np

from ._ffi import EcValue, check, lib

# CellType (src/ctype.rs:11-20; order of with_ct!, src/lib.rs:89-98)
UInt8, UInt16, UInt32, UInt64, Int8, Int16, Int32, Int64, Float32, Float64 = range(10)
CELL_TYPES = list(range(10))
CT_NAMES = ["UInt8", "UInt16", "UInt32", "UInt64", "Int8", "Int16", "Int32", "Int64", "Float32", "Float64"]
NP_DTYPES = [np.dtype(d) for d in (np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int16, np.int32,
                                   np.int64, np.float32, np.float64)]
ADD, SUB, MUL, DIV = range(4)

_stream: Optional[int] = None  # hipStream_t handle; None = default stream


def init(device: int = 0) -> None:
    check(lib().ec_init(device))


def set_stream(handle: Optional[int]) -> None:
    """Route subsequent launches to `handle` (e.g. torch.cuda.current_stream().cuda_stream)."""
    global _stream
    _stream = handle


def stream() -> Optional[int]:
    return _stream


def synchronize() -> None:
    check(lib().ec_stream_sync(_stream))


def cell_type_of(dtype) -> int:
    dt = np.dtype(dtype)
    for i, d in enumerate(NP_DTYPES):
        if d == dt:
            return i
    raise TypeError(f"no CellType for dtype {dt}")  # e.g. isize is not CellEncoding (encoding.rs:5-8)


def union(a: int, b: int) -> int:
    return lib().ec_union(a, b)


def can_fit_into(a: int, b: int) -> bool:
    return bool(lib().ec_can_fit_into(a, b))


class ParseError(ValueError):
    """Error::ParseError (src/error.rs, raised by `CellType::from_str`, src/ctype.rs:29-44)."""


def cell_type_to_string(ct: int) -> str:  # Display == Debug (ctype.rs:22-27)
    return CT_NAMES[ct]


def cell_type_from_str(s: str) -> int:  # ctype.rs:29-44
    for ct in CELL_TYPES:
        if CT_NAMES[ct] == s:
            return ct
    raise ParseError(f"Unable to parse '{s}' as CellType")


def is_integral(ct: int) -> bool:  # ctype.rs:55-69
    return NP_DTYPES[ct].kind in "ui"


def is_signed(ct: int) -> bool:  # ctype.rs:71-85
    return NP_DTYPES[ct].kind in "if"


def size_of(ct: int) -> int:  # ctype.rs:87-96
    return lib().ec_size_of(ct)


def min_value(ct: int) -> "CellValue":  # ctype.rs:158-167
    out = EcValue()
    check(lib().ec_min_value(ct, C.byref(out)))
    return CellValue.from_ec(out)


def max_value(ct: int) -> "CellValue":  # ctype.rs:170-179
    out = EcValue()
    check(lib().ec_max_value(ct, C.byref(out)))
    return CellValue.from_ec(out)


def zero(ct: int) -> "CellValue":  # ctype.rs:134-144
    return CellValue(ct, 0)


def one(ct: int) -> "CellValue":  # ctype.rs:146-156
    return CellValue(ct, 1)


# --------------------------------------------------------------------------- CellValue
def rust_debug(x) -> str:
    """`format!("{:?}", x)` of a Rust primitive: integers plain; floats as the shortest digits that round-trip
    for their width, decimal with at least one fractional digit for 1e-4 <= |x| < 1e16 (and zero), scientific
    (`1e16`, `1.5e-7`) outside, `NaN` / `inf` / `-inf`; bools `true` / `false`."""
    if isinstance(x, (bool, np.bool_)):
        return "true" if x else "false"
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    x = x if isinstance(x, np.floating) else np.float64(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    ax = abs(float(x))
    if ax == 0.0 or 1e-4 <= ax < 1e16:
        return np.format_float_positional(x, unique=True, trim="0")
    mant, exp = np.format_float_scientific(x, unique=True, trim="-").split("e")
    return f"{mant}e{int(exp)}"


def elided(values) -> str:
    """`Elided` (src/lib.rs:165-192): more than 10 items render as the first five, `, ... `, the last five."""
    items = [rust_debug(v) for v in values]
    if len(items) > 10:
        return ", ".join(items[:5]) + ", ... " + ", ".join(items[-5:])
    return ", ".join(items)


def _ends(n: int, fetch):
    """The cells a Debug rendering shows: all of them up to 10, else only the first and last five are downloaded."""
    if n <= 10:
        return list(fetch(0, n))
    return list(fetch(0, 5)) + [None] + list(fetch(n - 5, 5))


def _elided_ends(n: int, fetch) -> str:
    vals = _ends(n, fetch)
    if n > 10:
        return elided(vals[:5]) + ", ... " + elided(vals[6:])
    return elided(vals)


class CellValue:
    """Scalar with a run-time cell type (src/value.rs:12-20)."""

    __slots__ = ("ct", "value")

    def __init__(self, ct: int, value):
        self.ct = ct
        self.value = value if isinstance(value, np.generic) and value.dtype == NP_DTYPES[ct] else \
            np.array([value]).astype(NP_DTYPES[ct])[0]

    @staticmethod
    def new(x, ct: Optional[int] = None) -> "CellValue":
        """`x.into()`: numpy scalars keep their type; Python ints are i32, floats f64 (Rust literal defaults)."""
        if isinstance(x, CellValue):
            return x
        if ct is not None:
            return CellValue(ct, x)
        if isinstance(x, np.generic):
            return CellValue(cell_type_of(x.dtype), x)
        if isinstance(x, bool):
            raise TypeError("bool is not a CellEncoding")
        if isinstance(x, int):
            if not -2**31 <= x < 2**31:  # a Rust integer literal defaults to i32 and does not compile out of range
                raise OverflowError(f"literal out of range for `i32`: {x} (pass a numpy scalar of the intended type)")
            return CellValue(Int32, x)
        if isinstance(x, float):
            return CellValue(Float64, x)
        raise TypeError(f"cannot convert {type(x)} into CellValue")

    def cell_type(self) -> int:
        return self.ct

    def bits(self) -> int:
        return int.from_bytes(np.array([self.value], dtype=NP_DTYPES[self.ct]).tobytes(), "little")

    def to_ec(self) -> EcValue:
        v = EcValue()
        v.dtype = self.ct
        v.v.bits = self.bits()
        return v

    @staticmethod
    def from_ec(v: EcValue) -> "CellValue":
        n = NP_DTYPES[v.dtype].itemsize
        raw = int(v.v.bits & ((1 << (8 * n)) - 1)).to_bytes(n, "little")
        return CellValue(v.dtype, np.frombuffer(raw, dtype=NP_DTYPES[v.dtype])[0])

    def convert(self, ct: int) -> "CellValue":
        """CellValue::convert (value.rs:74-98): NarrowingError unless can_fit_into."""
        out = EcValue()
        src = self.to_ec()
        check(lib().ec_value_convert(C.byref(src), ct, C.byref(out)))
        return CellValue.from_ec(out)

    def get(self, ct: int):
        return self.convert(ct).value

    def to_f64(self) -> float:
        src = self.to_ec()
        return lib().ec_value_to_f64(C.byref(src))

    def _key(self, ct: int) -> int:
        c = self.convert(ct)
        if NP_DTYPES[ct].kind != "f":
            return int(c.value)
        n = NP_DTYPES[ct].itemsize * 8
        b = c.bits()
        if b >> (n - 1):
            b = b - (1 << n)  # as signed
            b ^= (1 << (n - 1)) - 1
        return b

    def cmp(self, other: "CellValue") -> int:
        """impl Ord for CellValue (value.rs:248-265): unify, ints natural, floats total_cmp."""
        ct = union(self.ct, other.ct)
        a, b = self._key(ct), other._key(ct)
        return (a > b) - (a < b)

    def __eq__(self, other):
        return self.cmp(CellValue.new(other)) == 0

    def __lt__(self, other):
        return self.cmp(CellValue.new(other)) < 0

    def __hash__(self):
        return hash((self.ct, self.bits()))

    def __gt__(self, other):
        return self.cmp(CellValue.new(other)) > 0

    def unify(self, other: "CellValue") -> tuple["CellValue", "CellValue"]:
        """value.rs:103-107: both converted to the union of their cell types."""
        ct = union(self.ct, other.ct)
        return self.convert(ct), other.convert(ct)

    @staticmethod
    def zero() -> "CellValue":  # num_traits::Zero (value.rs:166-170)
        return CellValue(UInt8, 0)

    @staticmethod
    def one() -> "CellValue":  # num_traits::One (value.rs:159-164)
        return CellValue(UInt8, 1)

    def is_zero(self) -> bool:  # value.rs:172-174
        return self.to_f64() == 0.0

    def _scalar_op(self, other, fn) -> "CellValue":
        """cv_bin_op! (value.rs:199-217): both sides as f64, result always Float64.  Scalar arithmetic stays on the
        host as in the reference (IEEE double arithmetic of the host CPU; the buffers' per-cell form runs on the GPU)."""
        a, b = np.float64(self.to_f64()), np.float64(CellValue.new(other).to_f64())
        with np.errstate(all="ignore"):
            return CellValue(Float64, fn(a, b))

    def __add__(self, other): return self._scalar_op(other, lambda a, b: a + b)
    def __sub__(self, other): return self._scalar_op(other, lambda a, b: a - b)
    def __mul__(self, other): return self._scalar_op(other, lambda a, b: a * b)
    def __truediv__(self, other): return self._scalar_op(other, lambda a, b: a / b)

    def __neg__(self) -> "CellValue":
        """impl Neg for CellValue (value.rs:224-240): u8 -> i16, u16 -> i32, u32/u64 -> f64, signed wrap, floats flip."""
        out_ct = lib().ec_neg_result_type(self.ct)
        dt = NP_DTYPES[out_ct]
        with np.errstate(all="ignore"):
            if dt.kind == "f":
                return CellValue(out_ct, -dt.type(self.value))
            return CellValue(out_ct, (np.array([self.value]).astype(dt) * dt.type(-1))[0])  # wrapping at MIN

    def __repr__(self):  # derived Debug: `Int32(37)`
        return f"{CT_NAMES[self.ct]}({rust_debug(self.value)})"


# --------------------------------------------------------------------------- device memory
class DeviceMem:
    """An HBM allocation (ec_alloc/ec_free), or a window into one (shards)."""

    _GLOBAL = object()

    def __init__(self, nbytes: int, _parent: "DeviceMem" = None, _offset: int = 0, stream=_GLOBAL):
        """`stream`: allocate (and later free) on this explicit stream — for callers that drive the ABI themselves on
        their own stream (e.g. one stream per host thread); default: the module's current stream (set_stream)."""
        self.nbytes = nbytes
        self._parent = _parent
        if _parent is not None:
            self.ptr = (_parent.ptr or 0) + _offset if nbytes else None
            self._owned = False
        else:
            p = C.c_void_p()
            self._explicit = stream is not DeviceMem._GLOBAL
            self._alloc_stream = stream if self._explicit else _stream  # the block goes back to the pool on this stream
            check(lib().ec_alloc_async(C.byref(p), nbytes, self._alloc_stream))  # stream-ordered pool: no per-op hipMalloc
            self.ptr = p.value
            self._owned = True

    def window(self, offset: int, nbytes: int) -> "DeviceMem":
        assert 0 <= offset and offset + nbytes <= self.nbytes
        return DeviceMem(nbytes, _parent=self, _offset=offset)

    def __del__(self):
        # The free is queued on the ALLOCATING stream, ordered (event + wait, inside ec_free_ordered) after everything
        # enqueued so far on the stream that is current now — the stream this buffer's last operator ran on when
        # the caller switched streams with set_stream() in between.  Freeing on "whatever stream is current" alone
        # would let the pool hand the block out again while kernels on the other stream still use it.
        if getattr(self, "_owned", False) and self.ptr:
            try:
                lib().ec_free_ordered(self.ptr, self._alloc_stream, self._alloc_stream if self._explicit else _stream)
            except Exception:
                pass
            self.ptr = None


def _upload(a: np.ndarray) -> DeviceMem:
    a = np.ascontiguousarray(a)
    mem = DeviceMem(a.nbytes)
    if a.nbytes:
        check(lib().ec_upload(mem.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, _stream))
    return mem


def _download(mem: DeviceMem, dtype, n: int) -> np.ndarray:
    out = np.empty(n, dtype=dtype)
    if out.nbytes:
        check(lib().ec_download(out.ctypes.data_as(C.c_void_p), mem.ptr, out.nbytes, _stream))
    return out


def _scalar(x) -> CellValue:
    return CellValue.new(x)


# --------------------------------------------------------------------------- CellBuffer
class CellBuffer:
    """Device-resident `CellBuffer` (src/buffer.rs:12-55): a cell type tag + n cells in HBM."""

    def __init__(self, ct: int, n: int, mem: DeviceMem):
        self.ct, self.n, self.mem = ct, n, mem

    # ---- constructors (BufferOps, src/lib.rs:104-163)
    @staticmethod
    def from_vec(data) -> "CellBuffer":
        a = np.ascontiguousarray(data)
        if a.dtype == np.dtype(np.int64) and not isinstance(data, np.ndarray):
            if a.size and (a.min() < -2**31 or a.max() >= 2**31):
                raise OverflowError("literal out of range for `i32` (pass a numpy array of the intended type)")
            a = a.astype(np.int32)  # Rust integer literals default to i32
        return CellBuffer(cell_type_of(a.dtype), a.size, _upload(a.ravel()))

    new = from_vec

    @staticmethod
    def from_values(values) -> "CellBuffer":
        """impl FromIterator<CellValue> for CellBuffer (src/buffer.rs:229-250): empty -> UInt8; otherwise the
        FIRST value's cell type, every value through `get::<T>().unwrap()` (a value that does not fit panics)."""
        vals = [CellValue.new(v) for v in values]
        if not vals:
            return CellBuffer.with_defaults(0, UInt8)
        ct = vals[0].ct
        return CellBuffer.from_vec(np.array([v.get(ct) for v in vals], dtype=NP_DTYPES[ct]))

    def __iter__(self):
        """impl IntoIterator for &CellBuffer (src/buffer.rs:278-305): yields CellValue; one download for the walk."""
        return (CellValue(self.ct, v) for v in self.to_numpy())

    @staticmethod
    def with_defaults(length: int, ct: int) -> "CellBuffer":
        return CellBuffer.fill(length, CellValue(ct, 0))

    @staticmethod
    def fill(length: int, value) -> "CellBuffer":
        v = _scalar(value)
        mem = DeviceMem(length * NP_DTYPES[v.ct].itemsize)
        ev = v.to_ec()
        check(lib().ec_fill(v.ct, mem.ptr, length, C.byref(ev), _stream))
        return CellBuffer(v.ct, length, mem)

    @staticmethod
    def fill_via(length: int, f: Callable[[int], object], dtype) -> "CellBuffer":
        return CellBuffer.from_vec(np.array([f(i) for i in range(length)], dtype=dtype))

    @staticmethod
    def empty(length: int, ct: int) -> "CellBuffer":
        return CellBuffer(ct, length, DeviceMem(length * NP_DTYPES[ct].itemsize))

    # ---- accessors
    def len(self) -> int:
        return self.n

    __len__ = len

    def is_empty(self) -> bool:
        return self.n == 0

    def cell_type(self) -> int:
        return self.ct

    def shard(self, cell_offset: int, cell_len: int) -> "CellBuffer":
        """Contiguous window (row-block shard) of this buffer; no copy."""
        sz = NP_DTYPES[self.ct].itemsize
        return CellBuffer(self.ct, cell_len, self.mem.window(cell_offset * sz, cell_len * sz))

    def get(self, index: int) -> CellValue:
        if not 0 <= index < self.n:
            raise IndexError(f"index out of bounds: the len is {self.n} but the index is {index}")  # Rust panics
        sz = NP_DTYPES[self.ct].itemsize
        return CellValue(self.ct, _download(self.mem.window(index * sz, sz), NP_DTYPES[self.ct], 1)[0])

    def put(self, index: int, value) -> None:
        v = _scalar(value).convert(self.ct)  # NarrowingError like buffer.rs:137
        if not 0 <= index < self.n:
            raise IndexError(f"index out of bounds: the len is {self.n} but the index is {index}")
        sz = NP_DTYPES[self.ct].itemsize
        a = np.array([v.value], dtype=NP_DTYPES[self.ct])
        check(lib().ec_upload(self.mem.window(index * sz, sz).ptr, a.ctypes.data_as(C.c_void_p), sz, _stream))

    def to_numpy(self) -> np.ndarray:
        return _download(self.mem, NP_DTYPES[self.ct], self.n)

    def extend(self, values) -> None:
        """impl Extend<C> for CellBuffer (src/buffer.rs:205-221): each item goes through num-traits'
        range-checked `to_<p>()` (value-based, unlike `convert`) and panics when it does not fit."""
        items = [_scalar(v) for v in values]
        dt = NP_DTYPES[self.ct]
        new = np.empty(len(items), dtype=dt)
        for i, v in enumerate(items):
            x = v.value
            if dt.kind in "ui":
                info = np.iinfo(dt)
                f = float(x)
                if f != f or not (info.min - 1 < f < info.max + 1):
                    raise OverflowError(f"called `Option::unwrap()` on a `None` value: {v!r} does not fit {CT_NAMES[self.ct]}")
                new[i] = int(x)  # truncation toward zero for float sources
            else:
                new[i] = dt.type(x)
        grown = DeviceMem((self.n + new.size) * dt.itemsize)
        if self.n:
            check(lib().ec_copy(grown.ptr, self.mem.ptr, self.n * dt.itemsize, _stream))
        if new.size:
            check(lib().ec_upload(grown.window(self.n * dt.itemsize, new.nbytes).ptr, new.ctypes.data_as(C.c_void_p), new.nbytes, _stream))
        self.mem, self.n = grown, self.n + new.size

    def clone(self) -> "CellBuffer":
        out = CellBuffer.empty(self.n, self.ct)
        check(lib().ec_copy(out.mem.ptr, self.mem.ptr, self.mem.nbytes, _stream))
        return out

    # ---- convert / to_vec / min_max (src/buffer.rs:150-185)
    def convert(self, ct: int) -> "CellBuffer":
        if ct == self.ct:
            return self.clone()
        if not can_fit_into(self.ct, ct):
            check(lib().ec_convert(self.ct, None, ct, None, 0, _stream))  # raises NarrowingError{src,dst}
        if self.n == 0:
            return CellBuffer.empty(0, UInt8)  # collect() of nothing (buffer.rs:233-234)
        out = CellBuffer.empty(self.n, ct)
        check(lib().ec_convert(self.ct, self.mem.ptr, ct, out.mem.ptr, self.n, _stream))
        return out

    def to_vec(self, ct: Optional[int] = None) -> np.ndarray:
        ct = self.ct if ct is None else ct
        r = self.convert(ct)
        if r.ct != ct:
            raise AssertionError("danger::cast: cell types differ (empty convert yields UInt8, buffer.rs:443-444)")
        return r.to_numpy()

    def min_max(self) -> tuple[CellValue, CellValue]:
        mn, mx = EcValue(), EcValue()
        check(lib().ec_min_max(self.ct, self.mem.ptr, None, self.n, C.byref(mn), C.byref(mx), _stream))
        return CellValue.from_ec(mn), CellValue.from_ec(mx)

    # ---- arithmetic (src/buffer.rs:321-371)
    def _binop(self, op: int, rhs) -> "CellBuffer":
        if isinstance(rhs, CellBuffer):
            n = min(self.n, rhs.n)  # zip (buffer.rs:327)
            if n == 0:
                return CellBuffer.empty(0, UInt8)
            out = CellBuffer.empty(n, Float64)
            check(lib().ec_binop(op, self.ct, self.mem.ptr, rhs.ct, rhs.mem.ptr, n, out.mem.ptr, _stream))
            return out
        v = _scalar(rhs).to_ec()  # RHS scalar (buffer.rs:346-352)
        if self.n == 0:
            return CellBuffer.empty(0, UInt8)
        out = CellBuffer.empty(self.n, Float64)
        check(lib().ec_binop_scalar(op, self.ct, self.mem.ptr, self.n, C.byref(v), out.mem.ptr, _stream))
        return out

    def __add__(self, rhs): return self._binop(ADD, rhs)
    def __sub__(self, rhs): return self._binop(SUB, rhs)
    def __mul__(self, rhs): return self._binop(MUL, rhs)
    def __truediv__(self, rhs): return self._binop(DIV, rhs)

    def __neg__(self) -> "CellBuffer":
        if self.n == 0:
            return CellBuffer.empty(0, UInt8)
        out = CellBuffer.empty(self.n, lib().ec_neg_result_type(self.ct))
        check(lib().ec_neg(self.ct, self.mem.ptr, self.n, out.mem.ptr, _stream))
        return out

    # ---- Ord / Eq (src/buffer.rs:373-436), decided on the device: first differing cell, no download
    def cmp(self, other: "CellBuffer") -> int:
        res = C.c_int32()
        check(lib().ec_buffer_cmp(self.ct, self.mem.ptr, self.n, other.ct, other.mem.ptr, other.n, C.byref(res), _stream))
        return res.value

    def __eq__(self, other):
        return isinstance(other, CellBuffer) and self.cmp(other) == 0

    def __lt__(self, other):
        return self.cmp(other) < 0

    def __gt__(self, other):
        return self.cmp(other) > 0

    __hash__ = None

    def __repr__(self):  # impl Debug for CellBuffer (src/buffer.rs:188-203)
        return f"{CT_NAMES[self.ct]}CellBuffer({_elided_ends(self.n, lambda o, k: self.shard(o, k).to_numpy())})"


# --------------------------------------------------------------------------- NoData
class NoData:
    """NoData<T> (src/masked/nodata.rs:7-17)."""

    NONE, DEFAULT, VALUE = range(3)

    def __init__(self, kind: int, value=None):
        self.kind, self._value = kind, value

    @staticmethod
    def none() -> "NoData":
        return NoData(NoData.NONE)

    @staticmethod
    def default() -> "NoData":
        return NoData(NoData.DEFAULT)

    @staticmethod
    def new(value) -> "NoData":
        return NoData(NoData.VALUE, value)

    def value(self, ct: int) -> Optional[CellValue]:
        """NoData::value (nodata.rs:23-40) for T = ct."""
        if self.kind == NoData.NONE:
            return None
        if self.kind == NoData.VALUE:
            v = self._value
            return v if isinstance(v, CellValue) and v.ct == ct else CellValue(ct, v.value if isinstance(v, CellValue) else v)
        out = EcValue()
        check(lib().ec_nodata_default(ct, C.byref(out)))
        return CellValue.from_ec(out)


# --------------------------------------------------------------------------- Mask
class Mask:
    """`Mask(Vec<bool>)` (src/masked/mask.rs:10-12): one byte per cell, 0 or 1, in HBM."""

    def __init__(self, n: int, mem: DeviceMem):
        self.n, self.mem = n, mem

    @staticmethod
    def new(values: Sequence[bool]) -> "Mask":
        a = np.ascontiguousarray(np.asarray(values).astype(bool).astype(np.uint8))
        return Mask(a.size, _upload(a))

    @staticmethod
    def fill(length: int, value: bool) -> "Mask":
        mem = DeviceMem(length)
        ev = CellValue(UInt8, 1 if value else 0).to_ec()
        check(lib().ec_fill(UInt8, mem.ptr, length, C.byref(ev), _stream))
        return Mask(length, mem)

    @staticmethod
    def fill_via(length: int, f: Callable[[int], bool]) -> "Mask":
        return Mask.new([bool(f(i)) for i in range(length)])

    @staticmethod
    def empty(length: int) -> "Mask":
        return Mask(length, DeviceMem(length))

    def len(self) -> int:
        return self.n

    __len__ = len

    def is_empty(self) -> bool:
        return self.n == 0

    def shard(self, cell_offset: int, cell_len: int) -> "Mask":
        return Mask(cell_len, self.mem.window(cell_offset, cell_len))

    def to_numpy(self) -> np.ndarray:
        return _download(self.mem, np.uint8, self.n)

    def get(self, index: int) -> bool:
        if not 0 <= index < self.n:
            raise IndexError(index)
        return bool(_download(self.mem.window(index, 1), np.uint8, 1)[0])

    def put(self, index: int, value: bool) -> None:
        if not 0 <= index < self.n:
            raise IndexError(index)
        a = np.array([1 if value else 0], dtype=np.uint8)
        check(lib().ec_upload(self.mem.window(index, 1).ptr, a.ctypes.data_as(C.c_void_p), 1, _stream))

    def __iter__(self):  # impl IntoIterator for Mask (mask.rs:171-178)
        return (bool(b) for b in self.to_numpy())

    def __getitem__(self, index: int) -> bool:  # impl Index<usize> for Mask (mask.rs:89-94)
        return self.get(index)

    def __setitem__(self, index: int, value: bool) -> None:  # impl IndexMut (mask.rs:96-100)
        self.put(index, value)

    def extend(self, values) -> None:  # impl Extend<bool> for Mask (mask.rs:83-87)
        new = np.asarray([bool(v) for v in values], dtype=np.uint8)
        grown = DeviceMem(self.n + new.size)
        if self.n:
            check(lib().ec_copy(grown.ptr, self.mem.ptr, self.n, _stream))
        if new.size:
            check(lib().ec_upload(grown.window(self.n, new.size).ptr, new.ctypes.data_as(C.c_void_p), new.size, _stream))
        self.mem, self.n = grown, self.n + new.size

    def clone(self) -> "Mask":
        out = Mask.empty(self.n)
        check(lib().ec_copy(out.mem.ptr, self.mem.ptr, self.n, _stream))
        return out

    def counts(self) -> tuple[int, int]:
        """(data, nodata) — mask.rs:72-80."""
        a, b = C.c_uint64(), C.c_uint64()
        check(lib().ec_mask_counts(self.mem.ptr, self.n, C.byref(a), C.byref(b), _stream))
        return a.value, b.value

    def all(self, value: bool) -> bool:
        t, f = self.counts()
        return (f == 0) if value else (t == 0)

    def __invert__(self) -> "Mask":  # Not for &Mask (mask.rs:111-116)
        out = Mask.empty(self.n)
        check(lib().ec_mask_not(self.mem.ptr, self.n, out.mem.ptr, _stream))
        return out

    def __and__(self, rhs: "Mask") -> "Mask":  # BitAnd for &Mask: zip -> shorter (mask.rs:129-140)
        n = min(self.n, rhs.n)
        out = Mask.empty(n)
        check(lib().ec_mask_and(self.mem.ptr, rhs.mem.ptr, n, out.mem.ptr, _stream))
        return out

    def __or__(self, rhs: "Mask") -> "Mask":  # BitOr for &Mask (mask.rs:153-163)
        n = min(self.n, rhs.n)
        out = Mask.empty(n)
        check(lib().ec_mask_or(self.mem.ptr, rhs.mem.ptr, n, out.mem.ptr, _stream))
        return out

    def __iand__(self, rhs: "Mask") -> "Mask":  # BitAnd for Mask (owned, in place; lhs length kept — mask.rs:118-127)
        check(lib().ec_mask_and(self.mem.ptr, rhs.mem.ptr, min(self.n, rhs.n), self.mem.ptr, _stream))
        return self

    def __ior__(self, rhs: "Mask") -> "Mask":  # mask.rs:142-151
        check(lib().ec_mask_or(self.mem.ptr, rhs.mem.ptr, min(self.n, rhs.n), self.mem.ptr, _stream))
        return self

    def cmp(self, other: "Mask") -> int:  # derived Ord on Vec<bool> (mask.rs:10)
        res = C.c_int32()
        check(lib().ec_buffer_cmp(UInt8, self.mem.ptr, self.n, UInt8, other.mem.ptr, other.n, C.byref(res), _stream))
        return res.value

    def __eq__(self, other):
        return isinstance(other, Mask) and self.cmp(other) == 0

    __hash__ = None

    def __repr__(self):  # impl Debug for Mask (src/masked/mask.rs:165-169)
        return f"Mask({_elided_ends(self.n, lambda o, k: self.shard(o, k).to_numpy().astype(bool))})"


# --------------------------------------------------------------------------- MaskedCellBuffer
class MaskedCellBuffer:
    """`MaskedCellBuffer(CellBuffer, Mask)` (src/masked/masked_buffer.rs:39-41)."""

    def __init__(self, buffer: CellBuffer, mask: Mask):
        if buffer.len() != mask.len():
            raise AssertionError("Mask and buffer must have the same length.")  # masked_buffer.rs:48-53
        self._buf, self._mask = buffer, mask

    new = None  # set below (classmethod-style alias needs the class)

    @staticmethod
    def from_vec(data) -> "MaskedCellBuffer":
        b = CellBuffer.from_vec(data)
        return MaskedCellBuffer(b, Mask.fill(b.len(), True))

    @staticmethod
    def from_iter(items, dtype=None) -> "MaskedCellBuffer":
        """FromIterator<C> (all valid) and FromIterator<(C, bool)> for MaskedCellBuffer (masked_buffer.rs:257-278)."""
        items = list(items)
        if items and isinstance(items[0], tuple):
            vals = np.array([p[0] for p in items]) if dtype is None else np.array([p[0] for p in items], dtype=dtype)
            if vals.dtype == np.dtype(np.int64) and dtype is None:
                vals = vals.astype(np.int32)
            return MaskedCellBuffer(CellBuffer.from_vec(vals), Mask.new([bool(p[1]) for p in items]))
        return MaskedCellBuffer.from_vec(items if dtype is None else np.array(items, dtype=dtype))

    def __iter__(self):
        """impl IntoIterator for &MaskedCellBuffer (masked_buffer.rs:289-318): yields (CellValue, bool)."""
        return ((CellValue(self.cell_type(), v), bool(m)) for v, m in zip(self._buf.to_numpy(), self._mask.to_numpy()))

    @staticmethod
    def from_buffer(b: CellBuffer) -> "MaskedCellBuffer":  # From<CellBuffer> (masked_buffer.rs:250-255)
        return MaskedCellBuffer(b, Mask.fill(b.len(), True))

    @staticmethod
    def from_vec_with_nodata(data, nodata: NoData) -> "MaskedCellBuffer":
        """masked_buffer.rs:62-71: mask[i] = !(data[i] == nodata) under total-order equality."""
        b = CellBuffer.from_vec(data)
        return MaskedCellBuffer(b, mask_from_nodata(b, nodata))

    @staticmethod
    def with_defaults(length: int, ct: int) -> "MaskedCellBuffer":
        return MaskedCellBuffer(CellBuffer.with_defaults(length, ct), Mask.fill(length, True))

    @staticmethod
    def fill(length: int, value) -> "MaskedCellBuffer":
        return MaskedCellBuffer(CellBuffer.fill(length, value), Mask.fill(length, True))

    @staticmethod
    def fill_via(length: int, f, dtype) -> "MaskedCellBuffer":
        return MaskedCellBuffer(CellBuffer.fill_via(length, f, dtype), Mask.fill(length, True))

    @staticmethod
    def fill_with_mask_via(length: int, mv: Callable[[int], tuple], dtype) -> "MaskedCellBuffer":
        pairs = [mv(i) for i in range(length)]
        return MaskedCellBuffer(CellBuffer.from_vec(np.array([p[0] for p in pairs], dtype=dtype)),
                                Mask.new([p[1] for p in pairs]))

    def buffer(self) -> CellBuffer:
        return self._buf

    def mask(self) -> Mask:
        return self._mask

    def len(self) -> int:
        return self._buf.len()

    __len__ = len

    def cell_type(self) -> int:
        return self._buf.cell_type()

    def shard(self, cell_offset: int, cell_len: int) -> "MaskedCellBuffer":
        return MaskedCellBuffer(self._buf.shard(cell_offset, cell_len), self._mask.shard(cell_offset, cell_len))

    def get(self, index: int) -> CellValue:
        return self._buf.get(index)

    def put(self, index: int, value) -> None:
        self._buf.put(index, value)

    def get_masked(self, index: int) -> Optional[CellValue]:
        return self._buf.get(index) if self._mask.get(index) else None

    def get_with_mask(self, index: int) -> tuple[CellValue, bool]:
        return self._buf.get(index), self._mask.get(index)

    def put_with_mask(self, index: int, value, mask: bool) -> None:
        self.put(index, value)
        self._mask.put(index, mask)

    def counts(self) -> tuple[int, int]:
        return self._mask.counts()

    def extend(self, pairs) -> None:  # impl Extend<(C, bool)> for MaskedCellBuffer (masked_buffer.rs:280-287)
        pairs = list(pairs)
        self._buf.extend([p[0] for p in pairs])
        self._mask.extend([p[1] for p in pairs])

    def convert(self, ct: int) -> "MaskedCellBuffer":
        return MaskedCellBuffer(self._buf.convert(ct), self._mask.clone())

    def to_vec(self, ct: Optional[int] = None) -> np.ndarray:
        return self._buf.to_vec(ct)

    def to_vec_with_nodata(self, ct: int, no_data: NoData) -> np.ndarray:
        """masked_buffer.rs:137-152: convert to T, then masked cells become no_data.value()."""
        conv = self._buf.convert(ct)
        if conv.ct != ct:
            raise AssertionError("danger::cast: cell types differ")
        nd = no_data.value(ct)
        if nd is None:
            return conv.to_numpy()
        out = CellBuffer.empty(conv.n, ct)
        ev = nd.to_ec()
        check(lib().ec_mask_select(ct, conv.mem.ptr, self._mask.mem.ptr, conv.n, C.byref(ev), out.mem.ptr, _stream))
        return out.to_numpy()

    def min_max(self) -> tuple[CellValue, CellValue]:
        mn, mx = EcValue(), EcValue()
        check(lib().ec_min_max(self.cell_type(), self._buf.mem.ptr, self._mask.mem.ptr, self.len(),
                               C.byref(mn), C.byref(mx), _stream))
        return CellValue.from_ec(mn), CellValue.from_ec(mx)

    # ---- arithmetic (src/masked/masked_buffer.rs:323-383)
    def _binop(self, op: int, rhs) -> "MaskedCellBuffer":
        if isinstance(rhs, MaskedCellBuffer):
            n = min(self.len(), rhs.len())
            if n == 0:
                return MaskedCellBuffer(CellBuffer.empty(0, UInt8), Mask.empty(0))
            out, om = CellBuffer.empty(n, Float64), Mask.empty(n)
            check(lib().ec_masked_binop(op, self._buf.ct, self._buf.mem.ptr, self._mask.mem.ptr,
                                        rhs._buf.ct, rhs._buf.mem.ptr, rhs._mask.mem.ptr, n,
                                        out.mem.ptr, om.mem.ptr, _stream))
            return MaskedCellBuffer(out, om)
        return MaskedCellBuffer(self._buf._binop(op, rhs), self._mask.clone())  # scalar: mask carried over (:353-364)

    def __add__(self, rhs): return self._binop(ADD, rhs)
    def __sub__(self, rhs): return self._binop(SUB, rhs)
    def __mul__(self, rhs): return self._binop(MUL, rhs)
    def __truediv__(self, rhs): return self._binop(DIV, rhs)

    def __neg__(self) -> "MaskedCellBuffer":
        return MaskedCellBuffer(-self._buf, self._mask.clone())

    def __eq__(self, other):  # derived PartialEq (masked_buffer.rs:39): buffer (all cells) and mask
        return isinstance(other, MaskedCellBuffer) and self._buf == other._buf and self._mask == other._mask

    def cmp(self, other: "MaskedCellBuffer") -> int:
        """derived PartialOrd (masked_buffer.rs:39): the buffer decides, the mask breaks ties; both on the device."""
        c = self._buf.cmp(other._buf)
        return c if c != 0 else self._mask.cmp(other._mask)

    def __lt__(self, other):
        return self.cmp(other) < 0

    def __gt__(self, other):
        return self.cmp(other) > 0

    __hash__ = None

    def __repr__(self):  # debug_tuple(buffer, mask) (src/masked/masked_buffer.rs:227-235)
        return f"{CT_NAMES[self.cell_type()]}MaskedCellBuffer({self._buf!r}, {self._mask!r})"


MaskedCellBuffer.new = staticmethod(lambda buffer, mask: MaskedCellBuffer(buffer, mask))


def mask_from_nodata(b: CellBuffer, nodata: NoData) -> Mask:
    m = Mask.empty(b.len())
    nd = nodata.value(b.ct)
    ev = nd.to_ec() if nd is not None else None
    check(lib().ec_mask_from_nodata(b.ct, b.mem.ptr, b.len(), C.byref(ev) if ev is not None else None,
                                    m.mem.ptr, _stream))
    return m
