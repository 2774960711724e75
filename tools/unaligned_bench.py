#!/usr/bin/env python3
"""Cost of cell offsets that are not 16-byte aligned (dev tool).

Times a few kernel families at 16384² cells on windows that start `off` cells into a slightly larger
allocation — what a row-block shard of a raster whose width is not a multiple of 16 cells looks like —
with the vector kernels (unaligned global access, default) and with the cell-wise kernels
(`ec_tune_set("unaligned_vector", 0)`).

    python tools/unaligned_bench.py [side] > gpurun_out/unaligned.md
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))

import torch  # noqa: E402

import erased_cells_hip as ec  # noqa: E402


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = side * side
    pad = 64
    torch.cuda.set_device(0)
    ec.init(0)
    L = ec.lib()
    stream = torch.cuda.current_stream().cuda_stream
    ec.set_stream(stream)
    chk = ec._ffi.check

    def synth(ct, seed):
        b = ec.CellBuffer.empty(n + pad, ct)
        chk(L.ec_synth_fill(ct, b.mem.ptr, n + pad, seed, 0, 1.0, 200.0, stream))
        return b

    a8, b16, f32 = synth(ec.UInt8, 1), synth(ec.UInt16, 2), synth(ec.Float32, 3)
    out = ec.CellBuffer.empty(n + pad, ec.Float64)
    m = ec.Mask.empty(n + pad)
    chk(L.ec_synth_mask(m.mem.ptr, n + pad, 7, 0, 30, stream))
    mo = ec.Mask.empty(n + pad)

    def timed(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def cases(off, out_off):
        l, r, f, o = a8.shard(off, n), b16.shard(off, n), f32.shard(off, n), out.shard(out_off, n)
        mm, mmo = m.shard(off, n), mo.shard(off, n)
        import ctypes as C
        mn, mx = ec._ffi.EcValue(), ec._ffi.EcValue()
        nd16 = ec.CellValue(ec.UInt16, 0).to_ec()
        return [
            ("binop Div UInt8/UInt16", 11, lambda: chk(L.ec_binop(ec.DIV, ec.UInt8, l.mem.ptr, ec.UInt16, r.mem.ptr, n, o.mem.ptr, stream))),
            ("binop Add UInt8+UInt16", 11, lambda: chk(L.ec_binop(ec.ADD, ec.UInt8, l.mem.ptr, ec.UInt16, r.mem.ptr, n, o.mem.ptr, stream))),
            ("convert Float32->Float64", 12, lambda: chk(L.ec_convert(ec.Float32, f.mem.ptr, ec.Float64, o.mem.ptr, n, stream))),
            ("convert UInt16->Float32", 6, lambda: chk(L.ec_convert(ec.UInt16, r.mem.ptr, ec.Float32, o.mem.ptr, n, stream))),
            ("convert UInt8->Float64", 9, lambda: chk(L.ec_convert(ec.UInt8, l.mem.ptr, ec.Float64, o.mem.ptr, n, stream))),
            ("mask_from_nodata UInt16", 3, lambda: chk(L.ec_mask_from_nodata(ec.UInt16, r.mem.ptr, n, C.byref(nd16), mmo.mem.ptr, stream))),
            ("mask_not", 2, lambda: chk(L.ec_mask_not(mm.mem.ptr, n, mmo.mem.ptr, stream))),
            ("min_max UInt16", 2, lambda: chk(L.ec_min_max(ec.UInt16, r.mem.ptr, None, n, C.byref(mn), C.byref(mx), stream))),
            ("fused NDVI UInt16", 12, lambda: ec.fused.ndvi(r, r.shard(0, n))),
        ]

    print(f"Unaligned windows, {side}x{side} = {n} cells, one MI355X, HIP-event timed, peak 8000 GB/s\n")
    print("| kernel | input offset (cells) | output offset (cells) | kernels | ms/launch | GB/s | frac of peak |")
    print("|---|---:|---:|---|---:|---:|---:|")
    for off, out_off in [(0, 0), (16, 0), (2, 0), (1, 0), (1, 1), (3, 1), (0, 1)]:
        for knob in (1, 0):
            if off == 0 and knob == 0:
                continue
            chk(L.ec_tune_set(b"unaligned_vector", knob))
            for name, bpc, fn in cases(off, out_off):
                ms = timed(fn)
                gbs = bpc * n / (ms * 1e-3) / 1e9
                print(f"| {name} | {off} | {out_off} | {'vector' if knob else 'cell-wise'} | {ms:.4f} | {gbs:.0f} | {gbs / 8000:.3f} |", flush=True)
    chk(L.ec_tune_set(b"unaligned_vector", 1))


if __name__ == "__main__":
    main()
