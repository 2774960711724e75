"""ctypes/numpy front-end to the parity oracle (oracle/libec_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product path.  See ec_oracle.h for
the pinning statement and the reference lines each function follows.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("EC_ORACLE_SO") or os.path.join(_HERE, "libec_oracle.so")  # EC_ORACLE_SO: e.g. the sanitizer build

# CellType discriminants (src/lib.rs:85-101 order, src/ctype.rs:16 repr(u8)).
U8, U16, U32, U64, I8, I16, I32, I64, F32, F64 = range(10)
NTYPES = 10
ADD, SUB, MUL, DIV = range(4)
OK, ERR_NARROWING, ERR_BADTYPE, ERR_NOMEM = range(4)
ND_NONE, ND_DEFAULT, ND_VALUE = range(3)

NP_DTYPES = [np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int16,
             np.int32, np.int64, np.float32, np.float64]
CT_NAMES = ["UInt8", "UInt16", "UInt32", "UInt64", "Int8", "Int16", "Int32",
            "Int64", "Float32", "Float64"]
_FIELDS = ["u8", "u16", "u32", "u64", "i8", "i16", "i32", "i64", "f32", "f64"]


def ct_of(arr: np.ndarray) -> int:
    for i, d in enumerate(NP_DTYPES):
        if arr.dtype == np.dtype(d):
            return i
    raise TypeError(f"no CellType for dtype {arr.dtype}")


class _Payload(C.Union):
    _fields_ = [("u8", C.c_uint8), ("u16", C.c_uint16), ("u32", C.c_uint32),
                ("u64", C.c_uint64), ("i8", C.c_int8), ("i16", C.c_int16),
                ("i32", C.c_int32), ("i64", C.c_int64), ("f32", C.c_float),
                ("f64", C.c_double), ("bits", C.c_uint64)]


class Value(C.Structure):
    """eco_value: 16-byte tagged scalar mirroring CellValue (src/value.rs:12-20)."""
    _fields_ = [("ct", C.c_uint8), ("pad_", C.c_uint8 * 7), ("v", _Payload)]

    @staticmethod
    def of(ct: int, x) -> "Value":
        r = Value()
        r.ct = ct
        r.v.bits = 0
        # go through numpy so NaN payloads / wrap-around survive unchanged
        a = np.array([x]).astype(NP_DTYPES[ct]) if not isinstance(x, np.generic) else np.array([x], dtype=NP_DTYPES[ct])
        raw = a.tobytes()
        r.v.bits = int.from_bytes(raw, "little")
        return r

    @staticmethod
    def from_bits(ct: int, bits: int) -> "Value":
        r = Value()
        r.ct = ct
        r.v.bits = bits
        return r

    def get(self):
        n = np.dtype(NP_DTYPES[self.ct]).itemsize
        raw = int(self.v.bits & ((1 << (8 * n)) - 1)).to_bytes(n, "little")
        return np.frombuffer(raw, dtype=NP_DTYPES[self.ct])[0]

    def bits(self) -> int:
        n = np.dtype(NP_DTYPES[self.ct]).itemsize
        return int(self.v.bits & ((1 << (8 * n)) - 1))

    def __repr__(self):
        return f"{CT_NAMES[self.ct]}({self.get()!r})"


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("ec_oracle.c", "ec_oracle.h", "Makefile")]
    if os.environ.get("EC_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        VP, SZ, I = C.c_void_p, C.c_size_t, C.c_int
        PV = C.POINTER(Value)
        L.eco_is_integral.argtypes = [I]
        L.eco_is_signed.argtypes = [I]
        L.eco_size_of.argtypes = [I]
        L.eco_size_of.restype = SZ
        L.eco_union.argtypes = [I, I]
        L.eco_can_fit_into.argtypes = [I, I]
        L.eco_min_value.argtypes = [I]
        L.eco_min_value.restype = Value
        L.eco_max_value.argtypes = [I]
        L.eco_max_value.restype = Value
        L.eco_value_convert.argtypes = [PV, I, PV]
        L.eco_value_binop.argtypes = [I, PV, PV]
        L.eco_value_binop.restype = Value
        L.eco_value_neg.argtypes = [PV]
        L.eco_value_neg.restype = Value
        L.eco_value_cmp.argtypes = [PV, PV]
        L.eco_value_eq.argtypes = [PV, PV]
        L.eco_value_to_f64.argtypes = [PV]
        L.eco_value_to_f64.restype = C.c_double
        L.eco_nodata_value.argtypes = [I, I, PV, PV]
        L.eco_binop.argtypes = [I, I, VP, SZ, I, VP, SZ, VP, C.POINTER(I), C.POINTER(SZ)]
        L.eco_binop_scalar.argtypes = [I, I, VP, SZ, PV, VP, C.POINTER(I), C.POINTER(SZ)]
        L.eco_neg.argtypes = [I, VP, SZ, VP, C.POINTER(I), C.POINTER(SZ)]
        L.eco_convert.argtypes = [I, VP, SZ, I, VP, C.POINTER(I), C.POINTER(SZ)]
        L.eco_min_max.argtypes = [I, VP, VP, SZ, PV, PV]
        L.eco_mask_from_nodata.argtypes = [I, VP, SZ, I, PV, VP]
        L.eco_mask_select.argtypes = [I, VP, VP, SZ, I, PV, VP]
        L.eco_mask_and.argtypes = [VP, SZ, VP, SZ, VP, C.POINTER(SZ)]
        L.eco_mask_and.restype = None
        L.eco_mask_or.argtypes = [VP, SZ, VP, SZ, VP, C.POINTER(SZ)]
        L.eco_mask_or.restype = None
        L.eco_mask_not.argtypes = [VP, SZ, VP]
        L.eco_mask_not.restype = None
        L.eco_mask_counts.argtypes = [VP, SZ, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.eco_mask_counts.restype = None
        L.eco_mask_all.argtypes = [VP, SZ, I]
        L.eco_buffer_cmp.argtypes = [I, VP, SZ, I, VP, SZ]
        L.ecof_set_threads.argtypes = [I]
        L.ecof_set_threads.restype = None
        L.ecof_binop.argtypes = [I, I, VP, I, VP, SZ, VP]
        L.ecof_binop_scalar.argtypes = [I, I, VP, SZ, PV, VP]
        L.ecof_neg.argtypes = [I, VP, SZ, VP, C.POINTER(I)]
        L.ecof_convert.argtypes = [I, VP, SZ, I, VP]
        L.ecof_min_max.argtypes = [I, VP, VP, SZ, PV, PV]
        L.ecof_mask_from_nodata.argtypes = [I, VP, SZ, PV, VP]
        L.ecof_mask_select.argtypes = [I, VP, VP, SZ, PV, VP]
        L.eco_splitmix64.argtypes = [C.c_uint64]
        L.eco_splitmix64.restype = C.c_uint64
        L.eco_fill_u8.argtypes = [VP, SZ, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
        L.eco_fill_u8.restype = None
        L.eco_fill_u16.argtypes = [VP, SZ, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
        L.eco_fill_u16.restype = None
        _lib = L
    return _lib


class NarrowingError(Exception):
    """Error::NarrowingError{src,dst} (src/error.rs:14-15)."""

    def __init__(self, src: int, dst: int):
        super().__init__(f"Invalid narrowing from cell-type {CT_NAMES[src]} to {CT_NAMES[dst]}")
        self.src, self.dst = src, dst


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a)


# ---- lattice / scalar -------------------------------------------------
def union(a: int, b: int) -> int:
    return lib().eco_union(a, b)


def can_fit_into(a: int, b: int) -> bool:
    return bool(lib().eco_can_fit_into(a, b))


def min_value(ct: int) -> Value:
    return lib().eco_min_value(ct)


def max_value(ct: int) -> Value:
    return lib().eco_max_value(ct)


def value_convert(v: Value, ct: int) -> Value:
    out = Value()
    rc = lib().eco_value_convert(C.byref(v), ct, C.byref(out))
    if rc == ERR_NARROWING:
        raise NarrowingError(v.ct, ct)
    assert rc == OK
    return out


def value_binop(op: int, l: Value, r: Value) -> Value:
    return lib().eco_value_binop(op, C.byref(l), C.byref(r))


def value_neg(v: Value) -> Value:
    return lib().eco_value_neg(C.byref(v))


def value_cmp(a: Value, b: Value) -> int:
    return lib().eco_value_cmp(C.byref(a), C.byref(b))


def value_eq(a: Value, b: Value) -> bool:
    return bool(lib().eco_value_eq(C.byref(a), C.byref(b)))


def value_to_f64(v: Value) -> float:
    return lib().eco_value_to_f64(C.byref(v))


def nodata_value(kind: int, ct: int, given: Value | None = None) -> Value | None:
    out = Value()
    g = given if given is not None else Value()
    return out if lib().eco_nodata_value(kind, ct, C.byref(g), C.byref(out)) else None


# ---- reference-shaped buffer ops --------------------------------------
def _finish(out: np.ndarray, out_ct: C.c_int, out_len: C.c_size_t) -> np.ndarray:
    ct, n = out_ct.value, out_len.value
    dt = np.dtype(NP_DTYPES[ct])
    return out.view(np.uint8)[: n * dt.itemsize].view(dt).copy()


def binop(op: int, l: np.ndarray, r: np.ndarray) -> np.ndarray:
    l, r = _c(l), _c(r)
    n = min(l.size, r.size)
    out = np.empty(max(n, 1), dtype=np.float64)
    oct_, olen = C.c_int(), C.c_size_t()
    rc = lib().eco_binop(op, ct_of(l), _p(l), l.size, ct_of(r), _p(r), r.size, _p(out), C.byref(oct_), C.byref(olen))
    assert rc == OK, rc
    return _finish(out, oct_, olen)


def binop_scalar(op: int, l: np.ndarray, rhs: Value) -> np.ndarray:
    l = _c(l)
    out = np.empty(max(l.size, 1), dtype=np.float64)
    oct_, olen = C.c_int(), C.c_size_t()
    rc = lib().eco_binop_scalar(op, ct_of(l), _p(l), l.size, C.byref(rhs), _p(out), C.byref(oct_), C.byref(olen))
    assert rc == OK, rc
    return _finish(out, oct_, olen)


def neg(a: np.ndarray) -> np.ndarray:
    a = _c(a)
    out = np.empty(max(a.size, 1), dtype=np.float64)
    oct_, olen = C.c_int(), C.c_size_t()
    rc = lib().eco_neg(ct_of(a), _p(a), a.size, _p(out), C.byref(oct_), C.byref(olen))
    assert rc == OK, rc
    return _finish(out, oct_, olen)


def convert(a: np.ndarray, dt: int) -> np.ndarray:
    a = _c(a)
    out = np.empty(max(a.size, 1), dtype=np.float64)
    oct_, olen = C.c_int(), C.c_size_t()
    rc = lib().eco_convert(ct_of(a), _p(a), a.size, dt, _p(out), C.byref(oct_), C.byref(olen))
    if rc == ERR_NARROWING:
        raise NarrowingError(ct_of(a), dt)
    assert rc == OK, rc
    return _finish(out, oct_, olen)


def min_max(a: np.ndarray, mask: np.ndarray | None = None) -> tuple[Value, Value]:
    a = _c(a)
    m = None if mask is None else _c(mask.astype(np.uint8))
    mn, mx = Value(), Value()
    rc = lib().eco_min_max(ct_of(a), _p(a), _p(m), a.size, C.byref(mn), C.byref(mx))
    assert rc == OK, rc
    return mn, mx


def mask_from_nodata(a: np.ndarray, nd_kind: int, nd: Value | None = None) -> np.ndarray:
    a = _c(a)
    mask = np.empty(a.size, dtype=np.uint8)
    g = nd if nd is not None else Value()
    rc = lib().eco_mask_from_nodata(ct_of(a), _p(a), a.size, nd_kind, C.byref(g), _p(mask))
    assert rc == OK, rc
    return mask


def mask_select(a: np.ndarray, mask: np.ndarray, nd_kind: int, nd: Value | None = None) -> np.ndarray:
    a, mask = _c(a), _c(mask.astype(np.uint8))
    out = np.empty_like(a)
    g = nd if nd is not None else Value()
    rc = lib().eco_mask_select(ct_of(a), _p(a), _p(mask), a.size, nd_kind, C.byref(g), _p(out))
    assert rc == OK, rc
    return out


def _mask2(fn, l, r):
    l, r = _c(l.astype(np.uint8)), _c(r.astype(np.uint8))
    out = np.empty(max(min(l.size, r.size), 1), dtype=np.uint8)
    olen = C.c_size_t()
    fn(_p(l), l.size, _p(r), r.size, _p(out), C.byref(olen))
    return out[: olen.value].copy()


def mask_and(l, r):
    return _mask2(lib().eco_mask_and, l, r)


def mask_or(l, r):
    return _mask2(lib().eco_mask_or, l, r)


def mask_not(m):
    m = _c(m.astype(np.uint8))
    out = np.empty_like(m)
    lib().eco_mask_not(_p(m), m.size, _p(out))
    return out


def mask_counts(m) -> tuple[int, int]:
    m = _c(m.astype(np.uint8))
    a, b = C.c_uint64(), C.c_uint64()
    lib().eco_mask_counts(_p(m), m.size, C.byref(a), C.byref(b))
    return a.value, b.value


def mask_all(m, value: bool) -> bool:
    m = _c(m.astype(np.uint8))
    return bool(lib().eco_mask_all(_p(m), m.size, int(value)))


def buffer_cmp(l: np.ndarray, r: np.ndarray) -> int:
    l, r = _c(l), _c(r)
    return lib().eco_buffer_cmp(ct_of(l), _p(l), l.size, ct_of(r), _p(r), r.size)


# ---- typed-loop forms --------------------------------------------------
def set_threads(n: int) -> None:
    lib().ecof_set_threads(n)


def f_binop(op: int, l: np.ndarray, r: np.ndarray, out: np.ndarray | None = None) -> np.ndarray:
    l, r = _c(l), _c(r)
    n = min(l.size, r.size)
    if out is None:
        out = np.empty(n, dtype=np.float64)
    rc = lib().ecof_binop(op, ct_of(l), _p(l), ct_of(r), _p(r), n, _p(out))
    assert rc == OK, rc
    return out


def f_binop_scalar(op: int, l: np.ndarray, rhs: Value) -> np.ndarray:
    l = _c(l)
    out = np.empty(l.size, dtype=np.float64)
    rc = lib().ecof_binop_scalar(op, ct_of(l), _p(l), l.size, C.byref(rhs), _p(out))
    assert rc == OK, rc
    return out


def f_neg(a: np.ndarray) -> np.ndarray:
    a = _c(a)
    out = np.empty(max(a.size, 1), dtype=np.float64)
    oct_ = C.c_int()
    rc = lib().ecof_neg(ct_of(a), _p(a), a.size, _p(out), C.byref(oct_))
    assert rc == OK, rc
    dt = np.dtype(NP_DTYPES[oct_.value])
    return out.view(np.uint8)[: a.size * dt.itemsize].view(dt).copy()


def f_convert(a: np.ndarray, dt: int) -> np.ndarray:
    a = _c(a)
    out = np.empty(a.size, dtype=NP_DTYPES[dt])
    rc = lib().ecof_convert(ct_of(a), _p(a), a.size, dt, _p(out))
    if rc == ERR_NARROWING:
        raise NarrowingError(ct_of(a), dt)
    assert rc == OK, rc
    return out


def f_min_max(a: np.ndarray, mask: np.ndarray | None = None) -> tuple[Value, Value]:
    a = _c(a)
    m = None if mask is None else _c(mask.astype(np.uint8))
    mn, mx = Value(), Value()
    rc = lib().ecof_min_max(ct_of(a), _p(a), _p(m), a.size, C.byref(mn), C.byref(mx))
    assert rc == OK, rc
    return mn, mx


def f_mask_from_nodata(a: np.ndarray, nd: Value | None) -> np.ndarray:
    a = _c(a)
    mask = np.empty(a.size, dtype=np.uint8)
    rc = lib().ecof_mask_from_nodata(ct_of(a), _p(a), a.size, None if nd is None else C.byref(nd), _p(mask))
    assert rc == OK, rc
    return mask


def f_mask_select(a: np.ndarray, mask: np.ndarray, nd: Value | None) -> np.ndarray:
    a, mask = _c(a), _c(mask.astype(np.uint8))
    out = np.empty_like(a)
    rc = lib().ecof_mask_select(ct_of(a), _p(a), _p(mask), a.size, None if nd is None else C.byref(nd), _p(out))
    assert rc == OK, rc
    return out


# ---- synthetic inputs -----------------------------------
def splitmix64(x: int) -> int:
    return lib().eco_splitmix64(x & 0xFFFFFFFFFFFFFFFF)


def fill_u8(n: int, seed: int, base: int = 0, lo: int = 0, hi: int = 255) -> np.ndarray:
    a = np.empty(n, dtype=np.uint8)
    lib().eco_fill_u8(_p(a), n, seed, base, lo, hi)
    return a


def fill_u16(n: int, seed: int, base: int = 0, lo: int = 0, hi: int = 65535) -> np.ndarray:
    a = np.empty(n, dtype=np.uint16)
    lib().eco_fill_u16(_p(a), n, seed, base, lo, hi)
    return a
