#!/usr/bin/env python3
"""Probe (dev tool): does a host range that was the source of a copy and is then UNMAPPED leave something behind in the HIP
runtime that bites when the addresses are mapped again?  A read-only file mapping goes through ec_host_expr (page-locking refused by
injection, or attempted for real), is unmapped, and new arrays are allocated and copied to / from many times.

    python tools/probe_unmap_reuse.py {refuse|real} [rounds]
"""
import gc
import mmap
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))
import numpy as np  # noqa: E402
import erased_cells_hip as ec  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "refuse"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    ec.init(0)
    L = ec.lib()
    L.ec_tune_set(b"inject_pin_refusal", 1 if mode == "refuse" else 0)
    n = 384 * 1000
    prog = [(ec.SUB, 0, 1, 0), (ec.ADD, 0, 1, 1), (ec.DIV, 4, 5, 0)]
    tmp = tempfile.mkdtemp()
    rng = np.random.default_rng(1)
    for r in range(rounds):
        a = rng.integers(1, 60000, n, dtype=np.uint16)
        b = rng.integers(0, 200, n, dtype=np.uint8)
        a.tofile(os.path.join(tmp, "a.bin")); b.tofile(os.path.join(tmp, "b.bin"))
        ra = np.memmap(os.path.join(tmp, "a.bin"), dtype=np.uint16, mode="r")
        rb = np.memmap(os.path.join(tmp, "b.bin"), dtype=np.uint8, mode="r")
        out = ec.fused.program_host([ra, rb], [], prog, chunk_cells=50_000)
        af, bf = a.astype(np.float64), b.astype(np.float64)
        assert np.array_equal(out, (af - bf) / (af + bf))
        del ra, rb
        gc.collect()  # the mappings are gone
        # new host memory, some of it at the old addresses: anonymous mappings of the same sizes, numpy arrays, copies both ways
        maps = [mmap.mmap(-1, 2 * n), mmap.mmap(-1, n)]
        for k in range(8):
            x = np.frombuffer(maps[k & 1], dtype=np.uint8)[:n]
            d = ec.CellBuffer.from_vec(x)
            y = d.to_numpy()
            assert np.array_equal(x, y)
            z = np.empty(n, np.float64)
            dz = ec.CellBuffer.from_vec(z)
            dz.to_numpy()
        del x, y, d
        for m_ in maps:
            try:
                m_.close()
            except BufferError:
                pass
    print(f"{mode}: {rounds} rounds of map / copy / unmap / remap / copy survived")


if __name__ == "__main__":
    main()
