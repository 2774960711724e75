"""Worker for tests/test_sharded_gloo.py: one rank of a world_size-N gloo job on CPU.

Exercises the N>1 host logic of the sharded path without a GPU: row-block shard
ranges (ec_shard_range), the {~key(min), key(max)} exchange (all_reduce MAX) and
its decode (ec_min_max_decode), and the counts exchange (all_reduce SUM).  The
per-shard reductions themselves come from the oracle here — on the GPU box the
same exchange is fed by ec_min_max_keys / ec_mask_counts_device.
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "erased-cells_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import erased_cells_hip as ec  # noqa: E402
from erased_cells_hip import sharded  # noqa: E402
from oracle import eco  # noqa: E402
from vectors import rand_cells, rand_mask  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    out = {}
    rows, cols = 37, 29  # rows not divisible by the world size
    for ct in range(eco.NTYPES):
        full = rand_cells(ct, rows * cols, 7)          # same seed on every rank = the whole raster
        mask = rand_mask(rows * cols, 8)
        off, ln = sharded.shard_range(rows, cols, rank, world)
        for use_mask in (False, True):
            m = mask[off:off + ln] if use_mask else None
            lmn, lmx = eco.f_min_max(full[off:off + ln], m)
            kmin = ec.CellValue(ct, lmn.get())._key(ct)
            kmax = ec.CellValue(ct, lmx.get())._key(ct)
            if ct == eco.U64:
                kmin, kmax = kmin - (1 << 63), kmax - (1 << 63)
            k0, k1 = sharded.allreduce_keys_host((~kmin, kmax))
            gmn, gmx = sharded.combine_min_max_keys(ct, (k0, k1))
            emn, emx = eco.f_min_max(full, mask if use_mask else None)
            assert (gmn.bits(), gmx.bits()) == (emn.bits(), emx.bits()), (ct, use_mask, rank)
        t = torch.tensor(eco.mask_counts(mask[off:off + ln]), dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        assert (int(t[0]), int(t[1])) == eco.mask_counts(mask)
    # shards tile the raster exactly
    t = torch.tensor([sharded.shard_range(rows, cols, rank, world)[1]], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    assert int(t[0]) == rows * cols
    out["ok"] = True
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
