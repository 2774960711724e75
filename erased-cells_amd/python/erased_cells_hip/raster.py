"""Raster ingest: the step before the hot path.

Mirrors the reference's GDAL adaptor — `RasterBandEx::{read_cells, read_cells_masked}`
(src/gdal/rasterband.rs:82-125), its 7-type subset (src/gdal/mod.rs:14-44) and the nodata
f64 -> NoData<T> conversion (src/gdal/mod.rs:49-70) — for the shape of raster the reference's
fixtures have: classic little-endian TIFF, one band, uncompressed strips, GDAL_NODATA in ASCII
tag 42113.  libgdal is not available in this environment.  Parsing is host work (numpy);
`read_cells*` upload once to HBM, and `*_rows` reads a row-block so a rank uploads only its shard.
"""
from __future__ import annotations

import math
import struct
from typing import Optional

import numpy as np

from . import buffer as B
from ._ffi import EcError, EC_ERR_ARG, EC_ERR_UNSUPPORTED_TYPE

_TYPE_SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8}
# (sample format, bits) -> dtype: TryFrom<GdalDataType> for CellType (src/gdal/mod.rs:30-44)
_GDAL_TYPES = {(1, 8): np.uint8, (1, 16): np.uint16, (1, 32): np.uint32, (2, 16): np.int16, (2, 32): np.int32,
               (3, 32): np.float32, (3, 64): np.float64}


class UnsupportedCellTypeError(EcError):
    """Error::UnsupportedCellTypeError (src/error.rs:16-17)."""

    def __init__(self, what: str):
        super().__init__(EC_ERR_UNSUPPORTED_TYPE, f"Unsupported cell-type {what}")


class NoDataConversionError(EcError):
    """Error::NoDataConversionError (src/error.rs:22-23)."""

    def __init__(self, nd: float, ty: str):
        super().__init__(EC_ERR_ARG, f"Unable to convert {nd} into NoData<{ty}>::Value")


def nodata_from_f64(ct: int, nd: Optional[float]) -> B.NoData:
    """TryFrom<GdalND> for NoData<T> (src/gdal/mod.rs:49-70): num-traits' range-checked f64.to_<T>()."""
    if nd is None:
        return B.NoData.none()
    dt = B.NP_DTYPES[ct]
    if dt.kind == "f":
        return B.NoData.new(B.CellValue(ct, dt.type(nd)))
    info = np.iinfo(dt)
    if math.isnan(nd) or not (info.min - 1 < nd < info.max + 1):
        raise NoDataConversionError(nd, dt.name)
    return B.NoData.new(B.CellValue(ct, dt.type(int(nd))))  # truncation toward zero


class RasterBand:
    """`Dataset::open(path)?.rasterband(1)` for the supported TIFF subset; cells stay on the host until read."""

    def __init__(self, cells: np.ndarray, no_data: Optional[float]):
        self.cells = cells            # [rows, cols]
        self.no_data = no_data

    @staticmethod
    def open(path: str) -> "RasterBand":
        d = open(path, "rb").read()
        if d[:4] != b"II*\x00":
            raise EcError(EC_ERR_ARG, f"{path}: not a little-endian classic TIFF")
        (off,) = struct.unpack_from("<I", d, 4)
        (n,) = struct.unpack_from("<H", d, off)
        tags = {}
        for i in range(n):
            tag, typ, cnt, _ = struct.unpack_from("<HHII", d, off + 2 + 12 * i)
            if typ not in _TYPE_SIZES:
                continue
            size = _TYPE_SIZES[typ] * cnt
            pos = off + 2 + 12 * i + 8
            if size > 4:
                (pos,) = struct.unpack_from("<I", d, pos)
            raw = d[pos:pos + size]
            if typ == 3:
                tags[tag] = list(struct.unpack(f"<{cnt}H", raw))
            elif typ == 4:
                tags[tag] = list(struct.unpack(f"<{cnt}I", raw))
            elif typ == 2:
                tags[tag] = raw.split(b"\0")[0].decode()
            else:
                tags[tag] = raw
        w, h = tags[256][0], tags[257][0]
        bits, comp, spp = tags.get(258, [1])[0], tags.get(259, [1])[0], tags.get(277, [1])[0]
        fmt = tags.get(339, [1])[0]
        if comp != 1 or spp != 1:
            raise EcError(EC_ERR_ARG, f"{path}: only uncompressed single-band TIFFs are supported")
        if (fmt, bits) not in _GDAL_TYPES:
            raise UnsupportedCellTypeError(f"sample format {fmt} with {bits} bits")
        dt = np.dtype(_GDAL_TYPES[(fmt, bits)])
        buf = b"".join(d[o:o + c] for o, c in zip(tags[273], tags[279]))
        if len(buf) != w * h * dt.itemsize:
            raise EcError(EC_ERR_ARG, f"{path}: strips do not cover the raster")
        cells = np.frombuffer(buf, dtype=dt.newbyteorder("<")).astype(dt).reshape(h, w)
        return RasterBand(cells, float(tags[42113]) if 42113 in tags else None)

    def size(self) -> tuple[int, int]:
        """raster_size(): (width, height)."""
        return self.cells.shape[1], self.cells.shape[0]

    def band_type(self) -> int:
        return B.cell_type_of(self.cells.dtype)

    def no_data_value(self) -> Optional[float]:
        return self.no_data

    def read_cells_rows(self, row0: int, nrows: int) -> B.CellBuffer:
        return B.CellBuffer.from_vec(self.cells[row0:row0 + nrows].ravel())

    def read_cells(self) -> B.CellBuffer:
        """RasterBandEx::read_cells (src/gdal/rasterband.rs:82-103), whole band."""
        return self.read_cells_rows(0, self.cells.shape[0])

    def read_cells_masked_rows(self, row0: int, nrows: int) -> B.MaskedCellBuffer:
        nd = nodata_from_f64(self.band_type(), self.no_data)
        return B.MaskedCellBuffer.from_vec_with_nodata(self.cells[row0:row0 + nrows].ravel(), nd)

    def read_cells_masked(self) -> B.MaskedCellBuffer:
        """RasterBandEx::read_cells_masked (src/gdal/rasterband.rs:104-125)."""
        return self.read_cells_masked_rows(0, self.cells.shape[0])
