#!/usr/bin/env python3
"""Dev tool: what an eager operator costs end to end when its result buffer is allocated per call
(the reference's `collect()` semantics) versus the kernel alone."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "erased-cells_amd", "python"))
import torch
import erased_cells_hip as ec
ec.init(0)
L = ec.lib(); chk = ec._ffi.check
for side in (2048, 8192, 16384):
    n = side * side
    a, b = ec.CellBuffer.empty(n, ec.UInt8), ec.CellBuffer.empty(n, ec.UInt16)
    chk(L.ec_synth_fill(ec.UInt8, a.mem.ptr, n, 1, 0, 0.0, 255.0, None)); chk(L.ec_synth_fill(ec.UInt16, b.mem.ptr, n, 2, 0, 1.0, 65535.0, None))
    out = ec.CellBuffer.empty(n, ec.Float64)
    for _ in range(60): chk(L.ec_binop(ec.DIV, ec.UInt8, a.mem.ptr, ec.UInt16, b.mem.ptr, n, out.mem.ptr, None))
    ec.synchronize()
    t = time.perf_counter()
    for _ in range(50): chk(L.ec_binop(ec.DIV, ec.UInt8, a.mem.ptr, ec.UInt16, b.mem.ptr, n, out.mem.ptr, None))
    ec.synchronize(); k = (time.perf_counter() - t) / 50
    r = a / b; del r; ec.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        r = a / b      # allocates the f64 result, frees the previous one
    ec.synchronize(); e = (time.perf_counter() - t) / 50
    print(f"side {side}: kernel only {k*1e3:.3f} ms   eager with per-op alloc/free {e*1e3:.3f} ms   overhead {(e-k)*1e3:.3f} ms")
