"""Host memory in, host memory out (`ec_host_expr`): the u8 / u16 -> f64 divide and EVI over 16384^2 cells whose operands and
result live in host memory — page-locked buffers, pageable buffers (registered for the call), and the naive path
(from_vec, operator, to_vec), next to the same operators on resident data.

    python tools/host_pipe_bench.py [side]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "erased-cells_amd", "python"))
import erased_cells_hip as ec  # noqa: E402

S, R, K = (lambda k: k), (lambda k: 4 + k), (lambda k: 8 + k)


def timed(fn, reps=5):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t)


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = side * side
    ec.init(0)
    P = ec.fused
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, n, dtype=np.uint8)
    b = rng.integers(1, 65536, n, dtype=np.uint16)
    c = rng.integers(1, 30000, n, dtype=np.uint16)
    div = [(ec.DIV, S(0), S(1), 0)]
    evi = [(ec.SUB, S(0), S(1), 0), (ec.MUL, R(0), K(0), 0), (ec.MUL, S(1), K(1), 1), (ec.ADD, S(0), R(1), 1),
           (ec.MUL, S(2), K(2), 2), (ec.SUB, R(1), R(2), 1), (ec.ADD, R(1), K(3), 1), (ec.DIV, R(0), R(1), 0)]
    ks = [2.5, 6.0, 7.5, 1.0]
    pa, pb, pc = P.pinned_empty(n, np.uint8), P.pinned_empty(n, np.uint16), P.pinned_empty(n, np.uint16)
    pa[:], pb[:], pc[:] = a, b, c
    pout = P.pinned_empty(n, np.float64)
    out = np.empty(n, np.float64)
    rows = []
    ec.lib().ec_tune_set(b"expr_jit", 0)  # the kernel is 1 % of the time here; keep the compiler thread out of the measurement

    def row(name, bpc, fn):
        t = timed(fn)
        rows.append((name, bpc, t, n / t / 1e9, bpc * n / t / 1e9))

    row("divide u8/u16, page-locked in and out (ec_host_expr)", 11, lambda: P.program_host([pa, pb], [], div, out=pout))
    row("divide u8/u16, pageable in and out (registered per call)", 11, lambda: P.program_host([a, b], [], div, out=out))
    row("divide u8/u16, naive: from_vec, operator, to_vec", 11, lambda: (ec.CellBuffer.from_vec(a) / ec.CellBuffer.from_vec(b)).to_numpy())
    row("EVI 3 x u16, page-locked in and out (ec_host_expr)", 14, lambda: P.program_host([pb, pc, pb], ks, evi, out=pout))
    row("EVI 3 x u16, pageable in and out (registered per call)", 14, lambda: P.program_host([b, c, b], ks, evi, out=out))
    row("EVI 3 x u16 masked (nodata 0 in every band -> mask AND -> nodata in the result), pageable", 14,
        lambda: P.program_host_masked([b, c, b], [0, 0, 0], ks, evi, out_nodata=-9999.0, out=out))
    for ch in (1 << 22, 1 << 23, 1 << 24, 1 << 26):
        row(f"divide u8/u16, page-locked, chunks of 2^{ch.bit_length() - 1} cells", 11, lambda ch=ch: P.program_host([pa, pb], [], div, out=pout, chunk_cells=ch))
    print(f"| host to host, {side}x{side} cells | link B/cell | seconds | Gcells/s | GB/s over the link (both directions) |")
    print("|---|---:|---:|---:|---:|")
    for name, bpc, t, g, gb in rows:
        print(f"| {name} | {bpc} | {t:.3f} | {g:.2f} | {gb:.1f} |")
    P.program_host([pb, pc, pb], ks, evi, out=pout)
    resident = P.program([ec.CellBuffer.from_vec(x[:100000]) for x in (b, c, b)], ks, evi).to_numpy()
    assert np.array_equal(pout[:100000].view(np.uint64), resident.view(np.uint64)), "streamed result differs from the resident one"


if __name__ == "__main__":
    main()
