#!/usr/bin/env python3
"""A/B of the leading-cell peel at odd input offsets (dev tool; knob `ec_tune_set("peel", 0|1|2)`).

0 = never peel, 1 = peel for 1-byte operands (the library's default), 2 = also for 2-byte operands.
Output buffers are fresh (16-byte aligned); inputs are windows `off` cells into larger buffers.

    python tools/peel_ab.py > gpurun_out/peel_ab.md
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
    n, pad = 16384 * 16384, 64

    def synth(ct, seed):
        b = ec.CellBuffer.empty(n + pad, ct)
        chk(L.ec_synth_fill(ct, b.mem.ptr, n + pad, seed, 0, 1.0, 200.0, stream))
        return b

    a8, b16, a16 = synth(ec.UInt8, 1), synth(ec.UInt16, 2), synth(ec.UInt16, 3)
    out = ec.CellBuffer.empty(n + pad, ec.Float64)

    def timed(fn, iters=40):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    names = {ec.ADD: "Add", ec.MUL: "Mul", ec.DIV: "Div"}
    print("Leading-cell peel A/B, 16384² cells, one MI355X, HIP-event timed through the C ABI, peak 8000 GB/s\n")
    print("| op | operands | input offset (cells) | peel | ms/launch | frac of peak |")
    print("|---|---|---:|---:|---:|---:|")
    for (lt, lb, ls), (rt, rb, rs), label in [((ec.UInt8, a8, 1), (ec.UInt16, b16, 2), "u8, u16"),
                                               ((ec.UInt16, a16, 2), (ec.UInt16, b16, 2), "u16, u16")]:
        for off in (0, 1):
            l, r, o = lb.shard(off, n), rb.shard(off, n), out.shard(0, n)
            for peel in ((1,) if off == 0 else (0, 1, 2)):
                chk(L.ec_tune_set(b"peel", peel))
                for op in (ec.ADD, ec.MUL, ec.DIV):
                    ms = timed(lambda: chk(L.ec_binop(op, lt, l.mem.ptr, rt, r.mem.ptr, n, o.mem.ptr, stream)))
                    frac = (ls + rs + 8) * n / (ms * 1e-3) / 1e9 / 8000
                    print(f"| {names[op]} | {label} | {off} | {peel} | {ms:.4f} | {frac:.3f} |", flush=True)
    chk(L.ec_tune_set(b"peel", 1))


if __name__ == "__main__":
    main()
