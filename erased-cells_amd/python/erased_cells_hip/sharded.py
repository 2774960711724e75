"""Row-block sharding of one raster across the ranks of a node.

One process per GPU (`torch.distributed`; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" on CPU for tests).  Shard g of G owns rows [g*R/G, (g+1)*R/G) of the
row-major raster (ec_shard_range); operands, masks and outputs use the same
split, so every element-wise kernel is local and needs no communication.  The
only exchanges are reductions of scalars:

  min_max  each rank reduces its shard to two order-preserving int64 keys
           {~key(min), key(max)} on its own GPU (ec_min_max_keys); one
           all_reduce(MAX) of that 16-byte tensor; decode (ec_min_max_decode).
  counts   all_reduce(SUM) of {n_true, n_false}.

The collective is latency-bound (16 B); link bandwidth is irrelevant.
"""
from __future__ import annotations

import ctypes as C

from . import buffer as B
from ._ffi import EcValue, check, lib


def shard_range(n_rows: int, n_cols: int, shard: int, n_shards: int) -> tuple[int, int]:
    """(cell_offset, cell_len) of a row-block shard."""
    off, ln = C.c_uint64(), C.c_uint64()
    check(lib().ec_shard_range(n_rows, n_cols, shard, n_shards, C.byref(off), C.byref(ln)))
    return off.value, ln.value


def combine_min_max_keys(ct: int, keys2) -> tuple[B.CellValue, B.CellValue]:
    """Decode all-reduced {~key(min), key(max)} into typed values."""
    arr = (C.c_int64 * 2)(int(keys2[0]), int(keys2[1]))
    mn, mx = EcValue(), EcValue()
    check(lib().ec_min_max_decode(ct, arr, C.byref(mn), C.byref(mx)))
    return B.CellValue.from_ec(mn), B.CellValue.from_ec(mx)


def sharded_min_max(local, group=None):
    """Global (min, max) of a raster whose local row-block is `local`
    (CellBuffer or MaskedCellBuffer on this rank's GPU).  Collective."""
    import torch
    import torch.distributed as dist

    buf = local.buffer() if isinstance(local, B.MaskedCellBuffer) else local
    mask_ptr = local.mask().mem.ptr if isinstance(local, B.MaskedCellBuffer) else None
    keys = torch.empty(2, dtype=torch.int64, device="cuda")
    B.set_stream(torch.cuda.current_stream().cuda_stream)  # same stream RCCL will order after
    check(lib().ec_min_max_keys(buf.ct, buf.mem.ptr, mask_ptr, buf.n, keys.data_ptr(), B.stream()))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(keys, op=dist.ReduceOp.MAX, group=group)
    k = keys.cpu()
    return combine_min_max_keys(buf.ct, (int(k[0]), int(k[1])))


def sharded_counts(local_mask: B.Mask, group=None) -> tuple[int, int]:
    """Global (data, nodata) counts of a row-sharded mask.  Collective."""
    import torch
    import torch.distributed as dist

    counts = torch.empty(2, dtype=torch.int64, device="cuda")
    B.set_stream(torch.cuda.current_stream().cuda_stream)
    check(lib().ec_mask_counts_device(local_mask.mem.ptr, local_mask.n, counts.data_ptr(), B.stream()))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    c = counts.cpu()
    return int(c[0]), int(c[1])


def allreduce_keys_host(keys2, group=None):
    """The same key exchange on host tensors (gloo): used by the CPU multi-process tests."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([int(keys2[0]), int(keys2[1])], dtype=torch.int64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t[0]), int(t[1])
