"""Cost model of the expression-program kernel (k_expr): time per launch of programs that differ in one thing at a time —
number of steps, operand kinds (stream / register / scalar), op — over three u16 streams of 16384^2 cells.

    python tools/expr_cost.py [side]      # prints a markdown table; needs a GPU
"""
import sys
import os

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "erased-cells_amd", "python"))
import erased_cells_hip as ec  # noqa: E402

ADD, SUB, MUL, DIV = ec.ADD, ec.SUB, ec.MUL, ec.DIV
S, R, K = (lambda k: k), (lambda k: 4 + k), (lambda k: 8 + k)


def main():
    jit = "--jit" in sys.argv  # every program compiled for itself (hiprtc) instead of interpreted
    if jit:
        sys.argv.remove("--jit")
    only = None
    if "--only" in sys.argv:  # one program, few launches: for counter passes under rocprofv3
        k = sys.argv.index("--only")
        only = sys.argv[k + 1]
        del sys.argv[k:k + 2]
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = side * side
    ec.init(0)
    L, E, C = ec.lib(), ec._ffi, __import__("ctypes")
    stream = torch.cuda.current_stream().cuda_stream
    ec.set_stream(stream)
    E.check(L.ec_tune_set(b"expr_jit", 2 if jit else 0))
    bands = [ec.CellBuffer.empty(n, ec.UInt16) for _ in range(3)]
    for i, b in enumerate(bands):
        E.check(L.ec_synth_fill(ec.UInt16, b.mem.ptr, n, 0x5EED0031 + i, 0, 2000.0, 30000.0, stream))
    out = ec.CellBuffer.empty(n, ec.Float64)
    sc = (E.EcValue * 4)(*[ec.CellValue.new(x).to_ec() for x in (2.5, 6.0, 7.5, 1.0)])
    dt = (C.c_uint8 * 3)(ec.UInt16, ec.UInt16, ec.UInt16)
    p = (C.c_void_p * 3)(*[b.mem.ptr for b in bands])

    def run(prog, reps=60):
        st = (E.EcExprStep * len(prog))(*[E.EcExprStep(*q) for q in prog])
        def go():
            E.check(L.ec_expr(dt, p, 3, sc, 4, st, len(prog), n, out.mem.ptr, stream))
        for _ in range(40 if reps > 5 else 2):
            go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            go()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    chain = lambda op, k, b: [(op, S(0), S(1), 0)] + [(op, R(0), b, 0)] * (k - 1)  # noqa: E731
    progs = {
        "1 step   s0+s1": chain(ADD, 1, S(2)),
        "2 steps  (s0+s1)+s2": chain(ADD, 2, S(2)),
        "4 steps  + stream": chain(ADD, 4, S(2)),
        "8 steps  + stream": chain(ADD, 8, S(2)),
        "16 steps + stream": chain(ADD, 16, S(2)),
        "8 steps  * stream": chain(MUL, 8, S(2)),
        "8 steps  + scalar": chain(ADD, 8, K(0)),
        "8 steps  + register (r0+r0)": chain(ADD, 8, R(0)),
        "8 steps  / stream": chain(DIV, 8, S(2)),
        "1 step   s0/s1": chain(DIV, 1, S(2)),
        "EVI (8 steps: 3 *scalar, 1 +scalar, 1 /)": [(SUB, S(0), S(1), 0), (MUL, R(0), K(0), 0), (MUL, S(1), K(1), 1), (ADD, S(0), R(1), 1),
                                                       (MUL, S(2), K(2), 2), (SUB, R(1), R(2), 1), (ADD, R(1), K(3), 1), (DIV, R(0), R(1), 0)],
    }
    print(("compiled (hiprtc)" if jit else "interpreter (k_expr)") + "\n")
    print(f"| program over u16 streams, {side}x{side} | B/cell (streams it reads + f64 out) | ms / launch | Gcells/s | fraction of 8 TB/s |")
    print("|---|---|---|---|---|")
    for name, prog in progs.items():
        if only and not name.startswith(only):
            continue
        ms = run(prog, reps=5 if only else 60)
        bpc = 8 + 2 * len({r for q in prog for r in q[1:3] if r < 4})
        print(f"| {name} | {bpc} | {ms:.3f} | {n / ms / 1e6:.1f} | {bpc * n / ms / 1e6 / 8000:.3f} |")


if __name__ == "__main__":
    main()
