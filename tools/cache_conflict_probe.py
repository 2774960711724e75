#!/usr/bin/env python3
"""Probe (dev tool): u8 ∘ u8 → f64 at 16384² loses when ONE of its two 256 MiB operands is loaded cacheable (profiles/r04/cache_plan_ab.md)
while u8 ∘ u16 gains.  Is it because the two equal, equally aligned operands fall on the same Infinity Cache sets at the same time?  The
one-set loop with the rhs at several byte offsets from its allocation's start, under each forced policy.

    python tools/cache_conflict_probe.py
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))
import torch  # noqa: E402
import erased_cells_hip as ec  # noqa: E402


def main():
    torch.cuda.set_device(0)
    ec.init(0)
    L, chk = ec.lib(), ec._ffi.check
    s = torch.cuda.current_stream().cuda_stream
    n = 16384 * 16384
    a = ec.CellBuffer.empty(n, ec.UInt8)
    big = ec.CellBuffer.empty(2 * n, ec.UInt8)
    out = ec.CellBuffer.empty(n, ec.Float64)
    chk(L.ec_synth_fill(ec.UInt8, a.mem.ptr, n, 1, 0, 1.0, 100.0, s))
    chk(L.ec_synth_fill(ec.UInt8, big.mem.ptr, 2 * n, 2, 0, 1.0, 100.0, s))
    print(f"lhs at {a.mem.ptr:#x}, rhs allocation at {big.mem.ptr:#x}, out at {out.mem.ptr:#x}")
    print("| op | rhs offset | none cacheable | lhs cacheable | rhs cacheable | both |\n|---|---:|---:|---:|---:|---:|")
    for op, opname in ((ec.ADD, "add"), (ec.DIV, "div")):
        for off in (0, 4096, 1 << 20, (1 << 20) + 4096, 64 << 20, (128 << 20) + 8192):
            row = []
            for force in (0, 1, 2, 3):
                chk(L.ec_tune_set(b"cache_force", force))
                rp = big.mem.ptr + off

                def run(k):
                    for _ in range(k):
                        chk(L.ec_binop(op, ec.UInt8, a.mem.ptr, ec.UInt8, rp, n, out.mem.ptr, s))
                run(80)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(60); e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 60
                row.append(10 * n / ms / 1e6 / 8000)
            print(f"| {opname} | {off} | " + " | ".join(f"{x:.4f}" for x in row) + " |")
    chk(L.ec_tune_set(b"cache_force", -1))


if __name__ == "__main__":
    main()
