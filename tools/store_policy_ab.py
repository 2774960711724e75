"""Do eager operator chains over rasters whose f64 temporaries fit the 256 MiB Infinity Cache gain from TEMPORAL stores (the next
operator's loads are cacheable under the load policy)?  The eager three-operator NDVI chain on u16 bands, nt stores (the
library) against a build with plain stores (make EXTRA=-DEC_NT_STORE=0), each library in its own process.

    python tools/store_policy_ab.py            # parent: runs both children, prints the table
"""
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
ALT = os.path.join(ROOT, "erased-cells_amd", "liberased_cells_hip_plainst.so")
SIDES = (1024, 2048, 3072, 4096, 5792, 8192)


def child():
    import torch
    sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))
    import erased_cells_hip as ec
    ec.init(0)
    L, chk = ec.lib(), ec._ffi.check
    stream = torch.cuda.current_stream().cuda_stream
    ec.set_stream(stream)
    for side in SIDES:
        n = side * side
        nir, red = ec.CellBuffer.empty(n, ec.UInt16), ec.CellBuffer.empty(n, ec.UInt16)
        chk(L.ec_synth_fill(ec.UInt16, nir.mem.ptr, n, 7, 0, 5000.0, 40000.0, stream))
        chk(L.ec_synth_fill(ec.UInt16, red.mem.ptr, n, 8, 0, 5000.0, 30000.0, stream))
        t1, t2, out = (ec.CellBuffer.empty(n, ec.Float64) for _ in range(3))

        def step():
            chk(L.ec_binop(ec.SUB, ec.UInt16, nir.mem.ptr, ec.UInt16, red.mem.ptr, n, t1.mem.ptr, stream))
            chk(L.ec_binop(ec.ADD, ec.UInt16, nir.mem.ptr, ec.UInt16, red.mem.ptr, n, t2.mem.ptr, stream))
            chk(L.ec_binop(ec.DIV, ec.Float64, t1.mem.ptr, ec.Float64, t2.mem.ptr, n, out.mem.ptr, stream))

        g = torch.cuda.CUDAGraph()  # replayed as a graph so that launch overhead does not hide the memory behaviour at small sizes
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        stream = cap.cuda_stream
        with torch.cuda.graph(g, stream=cap):
            for _ in range(10):
                step()
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        reps = max(3, int(2e9 / (n * 48 * 10)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(side, e0.elapsed_time(e1) / reps / 10, flush=True)


def main():
    if os.environ.get("EC_AB_CHILD"):
        return child()
    res = {}
    for name, lib in (("nt stores (library)", None), ("plain stores", ALT)):
        env = dict(os.environ, EC_AB_CHILD="1")
        if lib:
            env["EC_HIP_LIB"] = lib
        out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, check=True).stdout
        res[name] = {int(l.split()[0]): float(l.split()[1]) for l in out.strip().splitlines() if l and l[0].isdigit()}
    print("| side | f64 temporary MiB | eager NDVI chain ms, nt stores | plain stores | plain / nt |")
    print("|---:|---:|---:|---:|---:|")
    for side in SIDES:
        a, b = res["nt stores (library)"][side], res["plain stores"][side]
        print(f"| {side} | {side * side * 8 / 2**20:.0f} | {a:.4f} | {b:.4f} | {b / a:.3f} |")


if __name__ == "__main__":
    main()
