#!/usr/bin/env python3
"""Latency of the synchronous-result entry points on a fixture-sized buffer (186 x 169 cells) and on 16384² (dev tool).
Run twice: as is (results written straight into pinned host words) and with EC_NO_ZERO_COPY_RESULTS=1 (device scratch +
a queued device-to-host copy, the round-1 path)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))
import numpy as np  # noqa: E402

import erased_cells_hip as ec  # noqa: E402

ec.init(0)
L, chk = ec.lib(), ec._ffi.check
for n in (186 * 169, 16384 * 16384):
    a = ec.CellBuffer.empty(n, ec.UInt16)
    chk(L.ec_synth_fill(ec.UInt16, a.mem.ptr, n, 7, 0, 1.0, 65534.0, None))
    b = a.clone()
    m = ec.Mask.fill(n, True)
    mn, mx, t, f, idx = ec._ffi.EcValue(), ec._ffi.EcValue(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    calls = {"min_max": lambda: chk(L.ec_min_max(ec.UInt16, a.mem.ptr, None, n, C.byref(mn), C.byref(mx), None)),
             "mask_counts": lambda: chk(L.ec_mask_counts(m.mem.ptr, n, C.byref(t), C.byref(f), None)),
             "first_difference": lambda: chk(L.ec_first_difference(ec.UInt16, a.mem.ptr, b.mem.ptr, n, C.byref(idx), None))}
    for name, fn in calls.items():
        for spin in (0,):  # (round 4 tried polling the stream with hipStreamQuery for up to 200 µs before blocking: slower, removed)
            for _ in range(200):
                fn()
            reps = 2000 if n < 1 << 20 else 300
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            print(f"{'zero-copy' if not os.environ.get('EC_NO_ZERO_COPY_RESULTS') else 'copy     '}  n={n:>10}  {name:17s} sync_spin_us={spin:3d} {(time.perf_counter() - t0) / reps * 1e6:8.1f} us per call")
