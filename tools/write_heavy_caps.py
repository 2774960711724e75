#!/usr/bin/env python3
"""Occupancy caps for the WRITE-HEAVY kernels that still load (dev tool): a pure write runs best from two workgroups per CU, the 3:8 mix
needs full occupancy — where do buffer ∘ scalar (1-8 B read : 8 written), convert, neg and mask_select sit?  Each kernel over rotating
operand sets (every byte from HBM) under LDS reservations of 0 … 48 KiB per workgroup (knobs `scalar_lds_kb`, `map_lds_kb`).

    python tools/write_heavy_caps.py
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
    two = ec.CellValue.new(2.0).to_ec()
    SETS = 4

    def bufs(ct, seed):
        out = []
        for k in range(SETS):
            b = ec.CellBuffer.empty(n, ct)
            src = ct if ct in (ec.UInt8, ec.UInt16, ec.Float32, ec.Float64) else ec.UInt32
            chk(L.ec_synth_fill(src, b.mem.ptr, n if src == ct else n * ec.size_of(ct) // 4, seed + k, 0, 1.0, 200.0, s))
            out.append(b)
        return out

    def time_it(fn, reps=60):
        for i in range(30):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    caps = (0, 16, 24, 32, 48)
    print("| kernel (rotating sets, 16384²) | B/cell | " + " | ".join(f"{c} KiB" for c in caps) + " |\n|---|---:|" + "---:|" * len(caps))
    o64 = [ec.CellBuffer.empty(n, ec.Float64) for _ in range(2)]
    rows = []
    for ct, name in ((ec.UInt8, "u8"), (ec.UInt16, "u16"), (ec.Float32, "f32"), (ec.Float64, "f64")):
        src = bufs(ct, 10 * ct + 1)
        rows.append((f"binop_scalar Mul {name} ∘ 2.0", ec.size_of(ct) + 8, b"scalar_lds_kb",
                     lambda i, src=src, ct=ct: chk(L.ec_binop_scalar(ec.MUL, ct, src[i % SETS].mem.ptr, n, C.byref(two), o64[i & 1].mem.ptr, s))))
    u16 = bufs(ec.UInt16, 77)
    f32 = bufs(ec.Float32, 88)
    o32 = [ec.CellBuffer.empty(n, ec.Float32) for _ in range(2)]
    rows.append(("convert u16 → f64", 10, b"map_lds_kb", lambda i: chk(L.ec_convert(ec.UInt16, u16[i % SETS].mem.ptr, ec.Float64, o64[i & 1].mem.ptr, n, s))))
    rows.append(("convert u16 → f32", 6, b"map_lds_kb", lambda i: chk(L.ec_convert(ec.UInt16, u16[i % SETS].mem.ptr, ec.Float32, o32[i & 1].mem.ptr, n, s))))
    rows.append(("convert f32 → f64", 12, b"map_lds_kb", lambda i: chk(L.ec_convert(ec.Float32, f32[i % SETS].mem.ptr, ec.Float64, o64[i & 1].mem.ptr, n, s))))
    rows.append(("neg f32", 8, b"map_lds_kb", lambda i: chk(L.ec_neg(ec.Float32, f32[i % SETS].mem.ptr, n, o32[i & 1].mem.ptr, s))))
    f64 = bufs(ec.Float64, 99)
    f64b = bufs(ec.Float64, 111)
    u32 = bufs(ec.UInt32, 123)
    for (lt, lb, ln), (rt, rb, rn) in (((ec.Float64, f64, "f64"), (ec.Float64, f64b, "f64")), ((ec.Float64, f64, "f64"), (ec.Float32, f32, "f32")),
                                       ((ec.Float32, f32, "f32"), (ec.Float32, f32, "f32*")), ((ec.UInt32, u32, "u32"), (ec.Float32, f32, "f32")),
                                       ((ec.UInt16, u16, "u16"), (ec.Float64, f64, "f64"))):
        rows.append((f"binop Add {ln} ∘ {rn}", ec.size_of(lt) + ec.size_of(rt) + 8, b"binop_lds_kb",
                     lambda i, lt=lt, lb=lb, rt=rt, rb=rb: chk(L.ec_binop(ec.ADD, lt, lb[i % SETS].mem.ptr, rt, rb[(i + 1) % SETS].mem.ptr, n, o64[i & 1].mem.ptr, s))))
    for name, bpc, knob, fn in rows:
        cells = []
        for c in caps:
            chk(L.ec_tune_set(knob, c))
            cells.append(bpc * n / time_it(fn) / 1e6 / 8000)
        chk(L.ec_tune_set(knob, 0))
        print(f"| {name} | {bpc} | " + " | ".join(f"{x:.3f}" for x in cells) + " |")


if __name__ == "__main__":
    main()
