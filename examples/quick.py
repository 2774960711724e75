#!/usr/bin/env python3
"""examples/quick.rs of the reference, on the GPU: u8 buffer / u16 buffer * 0.5 -> f64 buffer."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "erased-cells_amd", "python"))
import erased_cells_hip as ec  # noqa: E402

ec.init(0)
buf1 = ec.CellBuffer.from_vec(np.array([1, 2, 3], np.uint8))
buf2 = ec.CellBuffer.from_vec(np.array([2, 4, 6], np.uint16))
result = buf1 / buf2 * 0.5
assert result == ec.CellBuffer.from_vec([0.25, 0.25, 0.25])
print(result, result.to_numpy())
