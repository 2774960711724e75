"""Fixture access for the tests: the reference's three Landsat TIFFs (tests/golden/*.tiff, data files
copied from testkit/data) read with Pillow — a reader that shares no code with the product, the one SURVEY.md
§8c derived the known-answer numbers with.  The product's own reader (erased_cells_hip.raster.RasterBand,
the mirror of src/gdal/rasterband.rs:82-125) is checked AGAINST this one in tests/test_raster_reader.py; nothing
the oracle is pinned with passes through product code."""
import numpy as np
from PIL import Image

GDAL_NODATA = 42113  # ASCII tag holding the band's nodata value (what GDALRasterBand::GetNoDataValue returns)


def read_tiff(path):
    """Return (cells[h, w], nodata_or_None); nodata as the f64 GDAL hands to src/gdal/mod.rs:49-70."""
    with Image.open(path) as im:
        cells = np.array(im)  # mode I;16 -> uint16, (rows, cols)
        tag = im.tag_v2.get(GDAL_NODATA)
    nodata = None if tag is None else float(str(tag).strip().strip("\x00"))
    return cells, nodata
