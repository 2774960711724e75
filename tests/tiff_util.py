"""Minimal baseline-TIFF strip reader for the reference's three Landsat fixtures
(tests/golden/*.tiff = /root/reference/testkit/data/*.tiff, data files only).

Stands in for `RasterBand::read_as` + `no_data_value()` (src/gdal/rasterband.rs:82-125)
for exactly the shape the fixtures have: little-endian, single band, uncompressed,
strip-organised, GDAL_NODATA in ASCII tag 42113.
"""
import struct

import numpy as np

_TYPE_SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 12: 8, 16: 8}


def read_tiff(path):
    """Return (cells[h, w], nodata_or_None)."""
    d = open(path, "rb").read()
    assert d[:4] == b"II*\x00", "little-endian classic TIFF only"
    (off,) = struct.unpack_from("<I", d, 4)
    (n,) = struct.unpack_from("<H", d, off)
    tags = {}
    for i in range(n):
        tag, typ, cnt, _ = struct.unpack_from("<HHII", d, off + 2 + 12 * i)
        size = _TYPE_SIZES[typ] * cnt
        pos = off + 2 + 12 * i + 8
        if size > 4:
            (pos,) = struct.unpack_from("<I", d, pos)
        raw = d[pos:pos + size]
        if typ == 3:
            val = list(struct.unpack(f"<{cnt}H", raw))
        elif typ == 4:
            val = list(struct.unpack(f"<{cnt}I", raw))
        elif typ == 2:
            val = raw.split(b"\0")[0].decode()
        else:
            val = raw
        tags[tag] = val
    w, h = tags[256][0], tags[257][0]
    bits, comp, spp = tags[258][0], tags[259][0], tags[277][0]
    fmt = tags.get(339, [1])[0]
    assert comp == 1 and spp == 1
    dt = {(1, 8): np.uint8, (1, 16): np.uint16, (1, 32): np.uint32, (2, 16): np.int16,
          (2, 32): np.int32, (3, 32): np.float32, (3, 64): np.float64}[(fmt, bits)]
    buf = b"".join(d[o:o + c] for o, c in zip(tags[273], tags[279]))
    cells = np.frombuffer(buf, dtype=np.dtype(dt).newbyteorder("<")).astype(dt).reshape(h, w)
    nodata = float(tags[42113]) if 42113 in tags else None
    return cells, nodata
