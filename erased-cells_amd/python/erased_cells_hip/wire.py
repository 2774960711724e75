"""serde wire shape of the core types, as JSON.

The reference derives `Serialize`/`Deserialize` for `CellType`, `CellValue`, `CellBuffer`, `Mask`,
`MaskedCellBuffer` and `NoData` (feature `serde`, on by default: src/ctype.rs:15, src/value.rs:16,
src/buffer.rs:51, src/masked/mask.rs:11, src/masked/masked_buffer.rs:40, src/masked/nodata.rs:8).  It has
no serialized fixture, no round-trip test and no format crate among its dependencies, so this module is
**parity unpinned**: it follows serde's documented mapping of derived types (externally tagged enums,
transparent newtype structs, tuple structs as sequences) as `serde_json` renders it:

    CellType::UInt8                    "UInt8"
    CellValue::UInt16(7)               {"UInt16": 7}
    CellBuffer::Float32(vec![1.5])     {"Float32": [1.5]}
    Mask(vec![true, false])            [true, false]
    MaskedCellBuffer(buffer, mask)     [{"UInt8": [1, 2]}, [true, false]]
    NoData::None / Default / Value(3)  "None" / "Default" / {"Value": 3}

Non-finite floats serialize as `null` (serde_json's rule) and, as there, do not deserialize.
Buffers cross the PCIe bus once in either direction; nothing here is on the per-cell path.
"""
from __future__ import annotations

import json
import math

import numpy as np

from .buffer import CELL_TYPES, CT_NAMES, NP_DTYPES, CellBuffer, CellValue, Mask, MaskedCellBuffer, NoData

_CT_BY_NAME = {CT_NAMES[ct]: ct for ct in CELL_TYPES}


class _F32(float):
    """A Float32 cell on its way to JSON: serde_json prints an f32 with the shortest digits that round-trip AS f32
    (0.1f32 -> `0.1`), not the digits of its f64 widening (0.10000000149011612)."""


def _num(x):
    if isinstance(x, float) and not math.isfinite(x):
        return None
    return x


def _f32_text(x: float) -> str:
    v = np.float32(x)
    ax = abs(float(v))
    if ax == 0.0 or 1e-5 <= ax < 1e16:  # ryu's plain-decimal range; always with a fractional part
        return np.format_float_positional(v, unique=True, trim="0")
    mant, exp = np.format_float_scientific(v, unique=True, trim="-").split("e")
    return f"{mant}e{int(exp)}"


def _emit(x) -> str:
    """Compact JSON of the small structure to_wire() builds (dict / list / str / bool / None / int / float / _F32)."""
    if isinstance(x, dict):
        return "{" + ",".join(json.dumps(k) + ":" + _emit(v) for k, v in x.items()) + "}"
    if isinstance(x, (list, tuple)):
        return "[" + ",".join(_emit(v) for v in x) + "]"
    if isinstance(x, _F32):
        return _f32_text(x)
    return json.dumps(x)


def cell_type_to_wire(ct: int) -> str:
    return CT_NAMES[ct]


def cell_type_from_wire(name: str) -> int:
    if name not in _CT_BY_NAME:
        raise ValueError(f"unknown variant `{name}`, expected one of {', '.join(_CT_BY_NAME)}")
    return _CT_BY_NAME[name]


def to_wire(x):
    """The JSON-ready structure serde_json would emit for `x`."""
    if isinstance(x, CellBuffer):
        vals = [_num(v) for v in x.to_numpy().tolist()]
        if x.cell_type() == CT_NAMES.index("Float32"):
            vals = [v if v is None else _F32(v) for v in vals]
        return {CT_NAMES[x.cell_type()]: vals}
    if isinstance(x, Mask):
        return [bool(b) for b in x.to_numpy().tolist()]
    if isinstance(x, MaskedCellBuffer):
        return [to_wire(x.buffer()), to_wire(x.mask())]
    if isinstance(x, CellValue):
        v = _num(x.value.item())
        return {CT_NAMES[x.cell_type()]: _F32(v) if (v is not None and x.cell_type() == CT_NAMES.index("Float32")) else v}
    if isinstance(x, NoData):
        if x.kind == NoData.NONE:
            return "None"
        if x.kind == NoData.DEFAULT:
            return "Default"
        v = x._value
        return {"Value": _num(v.value.item() if isinstance(v, CellValue) else np.asarray(v).item())}
    raise TypeError(f"no wire shape for {type(x).__name__}")


def _tagged(d, what: str):
    if not (isinstance(d, dict) and len(d) == 1):
        raise ValueError(f"invalid type for {what}: expected a map with a single variant key")
    (name, payload), = d.items()
    return cell_type_from_wire(name), payload


def _cells(ct: int, payload) -> np.ndarray:
    dt = NP_DTYPES[ct]
    if not isinstance(payload, list):
        raise ValueError("invalid type: expected a sequence")
    if any(v is None or isinstance(v, bool) for v in payload):
        raise ValueError(f"invalid type: null or boolean, expected {dt.name}")
    if dt.kind in "ui":
        info = np.iinfo(dt)
        for v in payload:
            if not isinstance(v, int) or not (info.min <= v <= info.max):
                raise ValueError(f"invalid value: {v!r}, expected {dt.name}")
    return np.array(payload, dtype=dt)


def buffer_from_wire(d) -> CellBuffer:
    ct, payload = _tagged(d, "CellBuffer")
    arr = _cells(ct, payload)
    return CellBuffer.from_vec(arr) if arr.size else CellBuffer.empty(0, ct)


def mask_from_wire(d) -> Mask:
    if not isinstance(d, list) or any(not isinstance(b, bool) for b in d):
        raise ValueError("invalid type for Mask: expected a sequence of booleans")
    return Mask.new(d)


def masked_from_wire(d) -> MaskedCellBuffer:
    if not (isinstance(d, list) and len(d) == 2):
        raise ValueError("invalid length for MaskedCellBuffer: expected a tuple of 2 elements")
    return MaskedCellBuffer(buffer_from_wire(d[0]), mask_from_wire(d[1]))


def value_from_wire(d) -> CellValue:
    ct, payload = _tagged(d, "CellValue")
    return CellValue(ct, _cells(ct, [payload])[0])


def nodata_from_wire(d, ct: int) -> NoData:
    if d == "None":
        return NoData.none()
    if d == "Default":
        return NoData.default()
    if isinstance(d, dict) and list(d) == ["Value"]:
        return NoData.new(CellValue(ct, _cells(ct, [d["Value"]])[0]))
    raise ValueError("unknown variant for NoData, expected one of None, Default, Value")


def dumps(x) -> str:
    return _emit(to_wire(x))


def loads_buffer(s: str) -> CellBuffer:
    return buffer_from_wire(json.loads(s))


def loads_masked(s: str) -> MaskedCellBuffer:
    return masked_from_wire(json.loads(s))
