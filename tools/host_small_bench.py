"""Host memory in, host memory out for SMALL rasters: ec_host_expr (which runs a small call as one chunk on the calling
thread's stream) against the same call forced through the pipeline, and against from_vec + operators + to_vec.

    python tools/host_small_bench.py
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "erased-cells_amd", "python"))
import erased_cells_hip as ec
ec.init(0)
P = ec.fused
S, R = (lambda k: k), (lambda k: 4 + k)
ndvi = [(ec.SUB, S(0), S(1), 0), (ec.ADD, S(0), S(1), 1), (ec.DIV, R(0), R(1), 0)]
rng = np.random.default_rng(0)
print("| cells | ec_host_expr ms | the same forced through the pipeline (chunk_cells = n) | from_vec + lazy (one pass) + to_vec | from_vec + eager operators + to_vec |")
print("|---:|---:|---:|---:|---:|")
for n in (31434, 1 << 18, 1 << 20, 1 << 21, 1 << 22, 1 << 24):
    a = rng.integers(1, 40000, n).astype(np.uint16); b = rng.integers(1, 30000, n).astype(np.uint16)
    out = np.empty(n)
    reps = 100 if n < (1 << 22) else 20
    def t(fn):
        for _ in range(5): fn()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        return (time.perf_counter() - t0) / reps * 1e3
    t_auto = t(lambda: P.program_host([a, b], [], ndvi, out=out))
    t_pipe = t(lambda: P.program_host([a, b], [], ndvi, out=out, chunk_cells=n))
    def lazy():
        da, db = ec.CellBuffer.from_vec(a), ec.CellBuffer.from_vec(b)
        return ((P.lazy(da) - db) / (P.lazy(da) + db)).eval().to_numpy()
    def naive():
        return ((ec.CellBuffer.from_vec(a) - ec.CellBuffer.from_vec(b)) / (ec.CellBuffer.from_vec(a) + ec.CellBuffer.from_vec(b))).to_numpy()
    print(f"| {n} | {t_auto:.3f} | {t_pipe:.3f} | {t(lazy):.3f} | {t(naive):.3f} |", flush=True)
