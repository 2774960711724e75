#!/usr/bin/env python3
"""A/B of the min_max launch shapes behind ec_tune_set("reduce_shape") at 16384² cells (dev tool):
0 = 512 threads x 8 loads (default), 1 = 512 x 16, 2 = 256 x 8 (8 workgroups/CU), 3 = 1024 x 8 (2/CU), 4 = 512 x 4.
Randomised order, several rounds, steady-state timing (ramp + >= 30 ms timed)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))
import torch  # noqa: E402

import erased_cells_hip as ec  # noqa: E402


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = side * side
    torch.cuda.set_device(0)
    ec.init(0)
    L, chk = ec.lib(), ec._ffi.check
    stream = torch.cuda.current_stream().cuda_stream
    ec.set_stream(stream)
    keys = torch.empty(2, dtype=torch.int64, device="cuda")
    bufs = {}
    for ct in (ec.UInt8, ec.UInt16, ec.Float32):
        b = ec.CellBuffer.empty(n, ct)
        chk(L.ec_synth_fill(ct, b.mem.ptr, n, 7 + ct, 0, 0.0, 200.0, stream))
        bufs[ct] = b
    m = ec.Mask.empty(n)
    chk(L.ec_synth_mask(m.mem.ptr, n, 11, 0, 30, stream))
    cases = [("min_max UInt8", ec.UInt8, None, 1), ("min_max UInt8 masked", ec.UInt8, m, 2), ("min_max UInt16", ec.UInt16, None, 2),
             ("min_max Float32 masked", ec.Float32, m, 5)]
    res = {}
    for rnd in range(5):
        order = [(c, sh) for c in range(len(cases)) for sh in range(5)]
        random.Random(rnd).shuffle(order)
        for c, sh in order:
            name, ct, mk, bpc = cases[c]
            chk(L.ec_tune_set(b"reduce_shape", sh))
            fn = lambda: chk(L.ec_min_max_keys(ct, bufs[ct].mem.ptr, mk.mem.ptr if mk else None, n, keys.data_ptr(), stream))
            for _ in range(300):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(400):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault((c, sh), []).append(e0.elapsed_time(e1) / 400)
    chk(L.ec_tune_set(b"reduce_shape", 0))
    print(f"min_max launch shapes, {side}x{side}, mean of 5 rounds x 400 launches (ms | fraction of 8 TB/s)\n")
    print("| case | 512x8 (default) | 512x16 | 256x8 | 1024x8 | 512x4 |")
    print("|---|---|---|---|---|---|")
    for c, (name, ct, mk, bpc) in enumerate(cases):
        cells = []
        for sh in range(5):
            ms = sum(res[(c, sh)]) / len(res[(c, sh)])
            cells.append(f"{ms:.4f} / {bpc * n / (ms * 1e-3) / 1e9 / 8000:.3f}")
        print(f"| {name} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
