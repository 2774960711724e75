"""Statistics of an index without its raster (`ec_expr_min_max`): NDVI and EVI over 16384^2 u16 bands — materialise + min_max
(the two-pass form, expr_jit = 0) against the compiled reduce kernel (expr_jit = 2); both answers compared.

    python tools/expr_stats_bench.py [side]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "erased-cells_amd", "python"))
import erased_cells_hip as ec  # noqa: E402

S, R, K = (lambda k: k), (lambda k: 4 + k), (lambda k: 8 + k)


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = side * side
    ec.init(0)
    L, E, P = ec.lib(), ec._ffi, ec.fused
    stream = torch.cuda.current_stream().cuda_stream
    ec.set_stream(stream)
    bands = [ec.CellBuffer.empty(n, ec.UInt16) for _ in range(3)]
    for i, b in enumerate(bands):
        E.check(L.ec_synth_fill(ec.UInt16, b.mem.ptr, n, 0x5EED0031 + i, 0, 2000.0, 30000.0, stream))
    ndvi = [(ec.SUB, S(0), S(1), 0), (ec.ADD, S(0), S(1), 1), (ec.DIV, R(0), R(1), 0)]
    evi = [(ec.SUB, S(0), S(1), 0), (ec.MUL, R(0), K(0), 0), (ec.MUL, S(1), K(1), 1), (ec.ADD, S(0), R(1), 1),
           (ec.MUL, S(2), K(2), 2), (ec.SUB, R(1), R(2), 1), (ec.ADD, R(1), K(3), 1), (ec.DIV, R(0), R(1), 0)]
    ks = [2.5, 6.0, 7.5, 1.0]
    print(f"| (min, max) of the index, {side}x{side} u16 bands | ms per call | bytes read + written per cell |")
    print("|---|---:|---:|")
    for name, streams, prog, sc, inb in (("NDVI", bands[:2], ndvi, [], 4), ("EVI", bands, evi, ks, 6)):
        answers = []
        for mode, how, bpc in ((0, "materialise (interpreter), then min_max", inb + 8 + 8), (2, "compiled reduce kernel", inb)):
            with P.jit(mode):
                for _ in range(5):
                    got = P.program_min_max(streams, sc, prog)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 20
                for _ in range(reps):
                    got = P.program_min_max(streams, sc, prog)
                ms = (time.perf_counter() - t0) / reps * 1e3
            answers.append((got[0].bits(), got[1].bits()))
            print(f"| {name}: {how} | {ms:.3f} | {bpc} |")
        assert answers[0] == answers[1], answers
    with P.jit(2):
        compiled_eager = P.program(bands[:2], [], ndvi)
    t0 = time.perf_counter()
    for _ in range(20):
        compiled_eager.min_max()
    print(f"| min_max of an f64 raster that is already there | {(time.perf_counter() - t0) / 20 * 1e3:.3f} | 8 |")


if __name__ == "__main__":
    main()
