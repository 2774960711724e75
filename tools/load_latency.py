#!/usr/bin/env python3
"""What the library's size costs at run time (dev tool): dlopen, ec_init, and the FIRST launch of one kernel from each
translation unit (HIP loads a translation unit's code object onto the device when its first kernel is launched), against the
second launch of the same kernel.  No torch, a fresh process.

    python tools/load_latency.py          (on an MI355X)
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))


def ms(f):
    t = time.perf_counter()
    r = f()
    return (time.perf_counter() - t) * 1e3, r


def split_dlopen(so):
    """In fresh processes: what a HIP program pays anyway (loading libamdhip64 and hipInit), and what loading THIS library adds
    once the runtime is up (registering its 21 code objects and their kernels with the runtime)."""
    import subprocess
    code = r"""
import ctypes as C, sys, time
t = time.perf_counter(); hip = C.CDLL("libamdhip64.so"); t_load = time.perf_counter() - t
t = time.perf_counter(); rc = hip.hipInit(0); n = C.c_int(0); hip.hipGetDeviceCount(C.byref(n)); t_init = time.perf_counter() - t
t = time.perf_counter(); lib = C.CDLL(sys.argv[1]); t_lib = time.perf_counter() - t
print(f"{t_load * 1e3:.1f} {t_init * 1e3:.1f} {t_lib * 1e3:.1f}")
"""
    r = subprocess.run([sys.executable, "-c", code, so], capture_output=True, text=True, timeout=120)
    a, b, c = (float(x) for x in r.stdout.split())
    print(f"in a fresh process: dlopen(libamdhip64.so) {a:.1f} ms, hipInit + hipGetDeviceCount {b:.1f} ms — any HIP program's cost; then dlopen of this "
          f"library (its static constructors register 21 code objects and every kernel in them) {c:.1f} ms")


def main():
    so = os.environ.get("EC_HIP_LIB") or os.path.join(ROOT, "erased-cells_amd", "liberased_cells_hip.so")
    print(f"library {so}: {os.path.getsize(so) / 1e6:.1f} MB")
    split_dlopen(so)
    t_import, _ = ms(lambda: __import__("erased_cells_hip"))
    import erased_cells_hip as ec
    t_dlopen, L = ms(ec.lib)
    t_init, _ = ms(lambda: ec.init(0))
    print(f"import of the Python mirror (numpy included) {t_import:.1f} ms; dlopen + binding every symbol {t_dlopen:.1f} ms; ec_init(0) {t_init:.1f} ms")
    chk = ec._ffi.check
    n = 1 << 20
    a, b = ec.CellBuffer.empty(n, ec.UInt8), ec.CellBuffer.empty(n, ec.UInt16)
    f, g = ec.CellBuffer.empty(n, ec.Float32), ec.CellBuffer.empty(n, ec.Float64)
    out = ec.CellBuffer.empty(n, ec.Float64)
    m1, m2 = ec.Mask.empty(n), ec.Mask.empty(n)
    keys = ec.CellBuffer.empty(2, ec.Int64)
    rows = []

    def first_and_second(name, call):
        def run():
            chk(call())
            chk(L.ec_stream_sync(None))
        t1, _ = ms(run)
        t2, _ = ms(run)
        rows.append((name, t1, t2))

    first_and_second("ec_synth_fill u8 (ec_abi.hip: map / reduce / generator kernels)", lambda: L.ec_synth_fill(ec.UInt8, a.mem.ptr, n, 1, 0, C.c_double(0), C.c_double(255), None))
    chk(L.ec_synth_fill(ec.UInt16, b.mem.ptr, n, 2, 0, C.c_double(1), C.c_double(65535), None))
    chk(L.ec_synth_fill(ec.Float32, f.mem.ptr, n, 3, 0, C.c_double(1), C.c_double(9), None))
    chk(L.ec_synth_mask(m1.mem.ptr, n, 4, 0, 30, None))
    chk(L.ec_synth_mask(m2.mem.ptr, n, 5, 0, 30, None))
    first_and_second("ec_binop Div u8 / u16 (ec_binop_div.hip)", lambda: L.ec_binop(ec.DIV, ec.UInt8, a.mem.ptr, ec.UInt16, b.mem.ptr, n, out.mem.ptr, None))
    first_and_second("ec_binop Add u8 + u16 (ec_binop_add.hip)", lambda: L.ec_binop(ec.ADD, ec.UInt8, a.mem.ptr, ec.UInt16, b.mem.ptr, n, out.mem.ptr, None))
    first_and_second("ec_min_max_keys u16 (ec_abi.hip, loaded above)", lambda: L.ec_min_max_keys(ec.UInt16, b.mem.ptr, None, n, keys.mem.ptr, None))
    dt4 = (C.c_uint8 * 4)(ec.UInt16, ec.UInt16, ec.UInt16, ec.UInt16)
    p4 = (C.c_void_p * 4)(b.mem.ptr, b.mem.ptr, b.mem.ptr, b.mem.ptr)
    first_and_second("ec_fused NDVI u16 (ec_fusedany_c2.hip)", lambda: L.ec_fused(ec.SUB, ec.DIV, ec.ADD, dt4, p4, None, n, out.mem.ptr, None))
    E = ec._ffi
    chk(L.ec_tune_set(b"expr_jit", 0))
    st = (E.EcExprStep * 2)(E.EcExprStep(ec.SUB, 0, 1, 0), E.EcExprStep(ec.DIV, 4, 0, 1))  # (a - b) / a: not in the catalogue
    dt2 = (C.c_uint8 * 2)(ec.UInt16, ec.Float32)
    p2 = (C.c_void_p * 2)(b.mem.ptr, f.mem.ptr)
    first_and_second("ec_expr interpreter, u16 and f32 streams (ec_expr_c2.hip)", lambda: L.ec_expr(dt2, p2, 2, None, 0, st, 2, n, out.mem.ptr, None))
    st3 = (E.EcExprStep * 3)(E.EcExprStep(ec.SUB, 0, 1, 0), E.EcExprStep(ec.ADD, 0, 1, 1), E.EcExprStep(ec.DIV, 4, 5, 0))
    dtn = (C.c_uint8 * 2)(ec.UInt16, ec.UInt16)
    c = ec.CellBuffer.empty(n, ec.UInt16)
    chk(L.ec_synth_fill(ec.UInt16, c.mem.ptr, n, 9, 0, C.c_double(1), C.c_double(65535), None))
    pn = (C.c_void_p * 2)(b.mem.ptr, c.mem.ptr)
    first_and_second("ec_expr NDVI u16, the built-in straight-line kernel (ec_expr_fixed.hip)", lambda: L.ec_expr(dtn, pn, 2, None, 0, st3, 3, n, out.mem.ptr, None))
    print("\n| first launch of a kernel from … | first call + sync, ms | second, ms |\n|---|---:|---:|")
    for name, t1, t2 in rows:
        print(f"| {name} | {t1:.2f} | {t2:.3f} |")
    print(f"\nsum of the first launches above: {sum(r[1] for r in rows):.1f} ms")


if __name__ == "__main__":
    main()
