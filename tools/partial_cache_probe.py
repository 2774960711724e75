# Is the 256 MiB u8 operand of the headline fully retained by the Infinity Cache, or would caching only a part of it do better?
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "erased-cells_amd", "python"))
import erased_cells_hip as ec
ec.init(0)
L, chk = ec.lib(), ec._ffi.check
stream = torch.cuda.current_stream().cuda_stream
ec.set_stream(stream)
side = 16384; n = side * side
a, b, out = ec.CellBuffer.empty(n, ec.UInt8), ec.CellBuffer.empty(n, ec.UInt16), ec.CellBuffer.empty(n, ec.Float64)
chk(L.ec_synth_fill(ec.UInt8, a.mem.ptr, n, 0x5EED0001, 0, 0.0, 255.0, stream))
chk(L.ec_synth_fill(ec.UInt16, b.mem.ptr, n, 0x5EED0002, 0, 1.0, 65535.0, stream))
def step(frac_rows_cached):
    rows_c = int(side * frac_rows_cached)
    nc = rows_c * side
    if nc:
        chk(L.ec_tune_set(b"mall_mb", 256))
        chk(L.ec_binop(ec.DIV, ec.UInt8, a.mem.ptr, ec.UInt16, b.mem.ptr, nc, out.mem.ptr, stream))
    if nc < n:
        chk(L.ec_tune_set(b"mall_mb", 0))
        chk(L.ec_binop(ec.DIV, ec.UInt8, a.mem.ptr + nc, ec.UInt16, b.mem.ptr + 2 * nc, n - nc, out.mem.ptr + 8 * nc, stream))
for frac in (1.0, 0.9375, 0.875, 0.75, 0.5, 0.0, 1.0):
    for _ in range(60): step(frac)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): step(frac)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    print(f"u8 rows cacheable {frac:6.4f}: {ms:.4f} ms  frac of peak {11 * n / ms / 1e6 / 8000:.4f}", flush=True)
chk(L.ec_tune_set(b"mall_mb", 256))
