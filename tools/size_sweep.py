#!/usr/bin/env python3
"""Throughput against raster size (dev tool): where the hot path turns from launch-bound to HBM-bound.

Times `ec_binop(Div, u8, u16)` (11 B/cell) and `ec_min_max_keys(u16)` (2 B/cell) through the C ABI for square
rasters from 256² to 32768² cells, back-to-back launches on one stream, HIP-event timed.

    python tools/size_sweep.py > gpurun_out/size_sweep.md
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))

import torch  # noqa: E402

import erased_cells_hip as ec  # noqa: E402


def main():
    torch.cuda.set_device(0)
    ec.init(0)
    L = ec.lib()
    stream = torch.cuda.current_stream().cuda_stream
    ec.set_stream(stream)
    chk = ec._ffi.check
    keys = torch.empty(2, dtype=torch.int64, device="cuda")

    def timed(fn, budget_ms):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        iters = max(20, min(20000, int(budget_ms / max(e0.elapsed_time(e1), 1e-3))))
        for _ in range(iters // 4):  # clock ramp
            fn()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3  # µs

    print("Raster-size sweep, one MI355X, back-to-back launches on one stream, HIP-event timed, peak 8000 GB/s\n")
    print("| side | cells | binop Div u8/u16 µs | Gcells/s | GB/s | frac | min_max u16 µs | Gcells/s | GB/s | frac |")
    print("|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for side in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
        n = side * side
        a, b = ec.CellBuffer.empty(n, ec.UInt8), ec.CellBuffer.empty(n, ec.UInt16)
        out = ec.CellBuffer.empty(n, ec.Float64)
        chk(L.ec_synth_fill(ec.UInt8, a.mem.ptr, n, 1, 0, 0.0, 255.0, stream))
        chk(L.ec_synth_fill(ec.UInt16, b.mem.ptr, n, 2, 0, 1.0, 65535.0, stream))
        t_div = timed(lambda: chk(L.ec_binop(ec.DIV, ec.UInt8, a.mem.ptr, ec.UInt16, b.mem.ptr, n, out.mem.ptr, stream)), 150.0)
        t_mm = timed(lambda: chk(L.ec_min_max_keys(ec.UInt16, b.mem.ptr, None, n, keys.data_ptr(), stream)), 150.0)
        g1, g2 = n / t_div / 1e3, n / t_mm / 1e3
        print(f"| {side} | {n} | {t_div:.2f} | {g1:.1f} | {g1 * 11:.0f} | {g1 * 11 / 8000:.3f} | {t_mm:.2f} | {g2:.1f} | {g2 * 2:.0f} | {g2 * 2 / 8000:.3f} |",
              flush=True)
        del a, b, out
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
