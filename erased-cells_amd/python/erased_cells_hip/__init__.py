"""erased_cells_hip — Python plumbing over liberased_cells_hip.so (MI355X / gfx950).

`_ffi`     ctypes binding of include/erased_cells.h (fails loudly if the .so is missing)
`buffer`   host mirror of CellBuffer / MaskedCellBuffer / Mask / NoData / CellValue
`fused`    operator chains and trees in one pass: `expr`, `ndvi`, `lazy()`, expression programs (`program`, interpreted or
           compiled for themselves), host memory in / host memory out (`program_host`, `program_host_masked`, `pinned_empty`)
`raster`   baseline-TIFF band reader (the reference's GDAL adaptor for the fixture shape)
`sharded`  row-block sharding across ranks + the RCCL all-reduce for min/max and counts; `ShardGroup`: one process, all GPUs
`wire`     serde/JSON shape of the core types (interop only; parity unpinned)
"""
from . import _ffi, fused, raster, wire
from ._ffi import EcError, NarrowingError, build, lib
from .buffer import (ADD, CELL_TYPES, CT_NAMES, DIV, MUL, NP_DTYPES, SUB, CellBuffer, CellValue, DeviceMem,
                     Float32, Float64, Int8, Int16, Int32, Int64, Mask, MaskedCellBuffer, NoData, UInt8,
                     UInt16, UInt32, UInt64, ParseError, can_fit_into, cell_type_from_str, cell_type_of,
                     cell_type_to_string, init, is_integral, is_signed, mask_from_nodata, max_value, min_value, one,
                     set_stream, size_of, stream, synchronize, union, zero)

__all__ = [n for n in dir() if not n.startswith("_")]
