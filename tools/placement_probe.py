#!/usr/bin/env python3
"""Probe (dev tool): how much does the RELATIVE PLACEMENT of a kernel's big streams matter?  f32 + f32 -> f64 and i64 + f64 -> f64 at
16384² (1-2 GiB per stream) came out at 0.76 on some boxes / runs and 0.82 on others with identical code.  The pool hands out blocks
of this size on 2 GiB boundaries, so all three streams walk the memory channels in lock-step; here the rhs and the output are moved
inside larger allocations by a list of byte offsets (rotating the lhs/rhs/out sets as bench.py does is NOT done: the point is the
placement, each configuration is a one-set loop with nothing cacheable).

    python tools/placement_probe.py
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
    L, chk = ec.lib(), ec._ffi.check
    s = torch.cuda.current_stream().cuda_stream
    n = 16384 * 16384
    pad = 300 << 20
    offs = [0, 4 << 10, 64 << 10, 1 << 20, (1 << 20) + (64 << 10), 16 << 20, (128 << 20) + (192 << 10), 256 << 20]
    for lt, rt, name in ((ec.Float32, ec.Float32, "f32 + f32"), (ec.Int64, ec.Float64, "i64 + f64"), (ec.UInt16, ec.UInt16, "u16 + u16")):
        ls, rs = ec.size_of(lt), ec.size_of(rt)
        a = ec.CellBuffer.empty(n, lt)
        bbig = ec.DeviceMem(n * rs + pad)
        obig = ec.DeviceMem(n * 8 + pad)
        chk(L.ec_fill(ec.UInt32, a.mem.ptr, n * ls // 4, __import__("ctypes").byref(ec.CellValue(ec.UInt32, 0x3f800000 if lt == ec.Float32 else 3).to_ec()), s))
        chk(L.ec_fill(ec.UInt32, bbig.ptr, (n * rs + pad) // 4, __import__("ctypes").byref(ec.CellValue(ec.UInt32, 0x3f800000 if rt == ec.Float32 else 0).to_ec()), s))
        bpc = ls + rs + 8
        print(f"\n{name} -> f64, {bpc} B/cell; lhs at {a.mem.ptr:#x}, rhs allocation at {bbig.ptr:#x}, out allocation at {obig.ptr:#x}")
        print("| rhs offset \\ out offset | " + " | ".join(f"{o >> 10} KiB" for o in offs) + " |\n|---|" + "---:|" * len(offs))
        for ro in offs:
            row = []
            for oo in offs:
                rp, op = bbig.ptr + ro, obig.ptr + oo

                def run(k):
                    for _ in range(k):
                        chk(L.ec_binop(ec.ADD, lt, a.mem.ptr, rt, rp, n, op, s))
                run(15)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(25); e1.record(); torch.cuda.synchronize()
                row.append(bpc * n / (e0.elapsed_time(e1) / 25) / 1e6 / 8000)
            print(f"| {ro >> 10} KiB | " + " | ".join(f"{x:.3f}" for x in row) + " |")
        del a, bbig, obig


if __name__ == "__main__":
    main()
