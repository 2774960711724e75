"""The product's TIFF strip reader (erased_cells_hip.raster.RasterBand — the ingest mirror of
src/gdal/rasterband.rs:82-125 for the fixture shape) against Pillow on the reference's three fixtures: same
cells, same shape, same cell type, same nodata.  The oracle's known-answer tests read the fixtures with Pillow
only (tests/tiff_util.py), so a bug in the product's reader cannot shift both sides of B.24-B.27."""
import os

import numpy as np
import pytest

from erased_cells_hip.raster import RasterBand
from tiff_util import read_tiff

FIXTURES = ["L8-Elkton-VA-B4.tiff", "L8-Elkton-VA-B5.tiff", "L8-Elkton-VA-B5-nd.tiff"]


@pytest.mark.parametrize("name", FIXTURES)
def test_raster_band_open_equals_pillow(golden_dir, name):
    path = os.path.join(golden_dir, name)
    cells, nodata = read_tiff(path)
    rb = RasterBand.open(path)
    assert rb.cells.dtype == cells.dtype == np.uint16 and rb.cells.shape == cells.shape == (169, 186)
    assert np.array_equal(rb.cells, cells)
    assert rb.no_data == nodata == 0.0  # all three carry GDAL_NODATA "0"; only -nd holds cells equal to it (4 of them)
    assert int((cells == 0).sum()) == (4 if name.endswith("-nd.tiff") else 0)


def test_oracle_tests_import_nothing_from_the_product():
    """tests/test_oracle_kat.py, test_oracle_forms.py and the fixture reader they use stay free of product code."""
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("test_oracle_kat.py", "test_oracle_forms.py", "tiff_util.py"):
        src = open(os.path.join(here, name)).read()
        assert not re.search(r"^\s*(from|import)\s+erased_cells_hip", src, re.M), name
