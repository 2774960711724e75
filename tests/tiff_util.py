"""Fixture access for the tests: the reference's three Landsat TIFFs (tests/golden/*.tiff, data files
copied from testkit/data) parsed by the package's host-side TIFF reader (erased_cells_hip.raster —
pure numpy, no GPU needed for parsing)."""
from erased_cells_hip.raster import RasterBand


def read_tiff(path):
    """Return (cells[h, w], nodata_or_None)."""
    rb = RasterBand.open(path)
    return rb.cells, rb.no_data
