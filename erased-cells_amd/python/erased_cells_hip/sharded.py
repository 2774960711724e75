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


# --------------------------------------------------------------------------- one process, all GPUs
class ShardedBuffer:
    """A raster cut into row-blocks, block i in device i's HBM (cells of one type, or a mask with ct = UInt8)."""

    def __init__(self, group: "ShardGroup", ct: int, ptrs, lens):
        self.group, self.ct, self.ptrs, self.lens = group, ct, ptrs, list(lens)

    def len(self) -> int:
        return sum(self.lens)

    def free(self) -> None:
        if self.ptrs is not None:
            check(lib().ec_sharded_free(self.group.handle, self.ptrs))
            self.ptrs = None


class ShardGroup:
    """`ec_shard_group`: one process driving the listed GPUs — per device a launch thread, a stream and a communicator
    of the clique inside the library.  `host_combine=True` folds the 16-byte reduction payloads on the host instead of
    over xGMI (no RCCL; lets one device be listed several times, e.g. to rehearse on a 1-GPU box)."""

    def __init__(self, devices, host_combine: bool = False, blocking_issue: bool = False):
        """`blocking_issue=True`: every element-wise call waits until all launch threads have issued (the form of rounds
        1-2); default: the calls are queued for the launch threads and return at once, a failure inside one is raised by
        the next `sync()` / reduction."""
        self.n = len(devices)
        self.handle = C.c_void_p()
        arr = (C.c_int32 * self.n)(*devices)
        flags = (1 if host_combine else 0) | (2 if blocking_issue else 0)
        check(lib().ec_shard_group_create(arr, self.n, flags, C.byref(self.handle)))

    def close(self) -> None:
        if self.handle:
            check(lib().ec_shard_group_destroy(self.handle))
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def shard(self, i: int):
        dev, st = C.c_int32(), C.c_void_p()
        check(lib().ec_shard_group_shard(self.handle, i, C.byref(dev), C.byref(st)))
        return dev.value, st.value

    def sync(self) -> None:
        """Everything queued so far has been issued and has finished; raises the first failure a queued call left behind."""
        check(lib().ec_shard_group_sync(self.handle))

    def stat(self, key: str) -> int:
        v = C.c_int64()
        check(lib().ec_shard_group_stat(self.handle, key.encode(), C.byref(v)))
        return v.value

    def foreach(self, fn) -> None:
        """fn(shard, device, stream) on every shard's own launch thread (its device current); any ABI call may be made
        there on `stream`.  An exception raised by fn is re-raised here."""
        from ._ffi import SHARD_FN, EC_ERR_ARG
        raised = []

        def tramp(shard, device, stream, _user):
            try:
                fn(shard, device, stream)
                return 0
            except Exception as e:  # noqa: BLE001 — carried across the C frame
                raised.append(e)
                return EC_ERR_ARG

        cb = SHARD_FN(tramp)
        st = lib().ec_shard_group_foreach(self.handle, cb, None)
        if raised:
            raise raised[0]
        check(st)

    def _sizes(self, lens, itemsize):
        return (C.c_size_t * self.n)(*[ln * itemsize for ln in lens])

    def alloc(self, ct: int, lens) -> ShardedBuffer:
        ptrs = (C.c_void_p * self.n)()
        check(lib().ec_sharded_alloc(self.handle, self._sizes(lens, B.NP_DTYPES[ct].itemsize), ptrs))
        return ShardedBuffer(self, ct, ptrs, lens)

    def scatter(self, a, n_rows: int, n_cols: int, ct: int = None) -> ShardedBuffer:
        """`From<Vec<T>>` of a row-major raster: contiguous row-blocks, block i to device i."""
        import numpy as np
        a = np.ascontiguousarray(a).ravel()
        assert a.size == n_rows * n_cols
        ct = B.cell_type_of(a.dtype) if ct is None else ct
        rng = [shard_range(n_rows, n_cols, g, self.n) for g in range(self.n)]
        out = self.alloc(ct, [r[1] for r in rng])
        sz = a.dtype.itemsize
        offs = (C.c_size_t * self.n)(*[r[0] * sz for r in rng])
        check(lib().ec_sharded_upload(self.handle, out.ptrs, a.ctypes.data_as(C.c_void_p), offs, self._sizes(out.lens, sz)))
        return out

    def gather(self, sb: ShardedBuffer):
        import numpy as np
        dt = B.NP_DTYPES[sb.ct]
        out = np.empty(sb.len(), dtype=dt)
        offs, acc = [], 0
        for ln in sb.lens:
            offs.append(acc * dt.itemsize)
            acc += ln
        check(lib().ec_sharded_download(self.handle, out.ctypes.data_as(C.c_void_p), sb.ptrs, (C.c_size_t * self.n)(*offs),
                                        self._sizes(sb.lens, dt.itemsize)))
        return out

    def binop(self, op: int, l: ShardedBuffer, r: ShardedBuffer) -> ShardedBuffer:
        assert l.lens == r.lens, "operands must be sharded identically"
        out = self.alloc(B.Float64, l.lens)
        check(lib().ec_sharded_binop(self.handle, op, l.ct, l.ptrs, r.ct, r.ptrs, (C.c_size_t * self.n)(*l.lens), out.ptrs))
        return out

    def program(self, streams, scalars, steps, masks=None):
        """An expression program (`fused.program`: steps `(op, a, b, dst)`) on every shard, fire-and-forget: `streams` up to four
        identically sharded buffers, `masks` (optional) one sharded mask per stream.  Returns the f64 result, or
        `(result, mask)` with `masks`."""
        from ._ffi import EcExprStep, PVP
        lens = streams[0].lens
        assert all(b.lens == lens for b in streams), "operands must be sharded identically"
        k = len(streams)
        dt = (C.c_uint8 * k)(*[b.ct for b in streams])
        p = (PVP * k)(*[C.cast(b.ptrs, PVP) for b in streams])
        m = (PVP * k)(*[C.cast(b.ptrs, PVP) for b in masks]) if masks is not None else None
        sc = (EcValue * max(1, len(scalars)))(*[B.CellValue.new(x).to_ec() for x in scalars])
        st = (EcExprStep * len(steps))(*[EcExprStep(*q) for q in steps])
        out = self.alloc(B.Float64, lens)
        om = self.alloc(B.UInt8, lens) if masks is not None else None
        check(lib().ec_sharded_expr(self.handle, dt, p, m, k, sc, len(scalars), st, len(steps), (C.c_size_t * self.n)(*lens), out.ptrs,
                                    om.ptrs if om is not None else None))
        return out if om is None else (out, om)

    def program_min_max(self, streams, scalars, steps, masks=None):
        """`(min, max)` of a program's result over the whole sharded raster without the raster (`ec_sharded_expr_min_max`)."""
        from ._ffi import EcExprStep, PVP
        lens = streams[0].lens
        assert all(b.lens == lens for b in streams), "operands must be sharded identically"
        k = len(streams)
        dt = (C.c_uint8 * k)(*[b.ct for b in streams])
        p = (PVP * k)(*[C.cast(b.ptrs, PVP) for b in streams])
        m = (PVP * k)(*[C.cast(b.ptrs, PVP) for b in masks]) if masks is not None else None
        sc = (EcValue * max(1, len(scalars)))(*[B.CellValue.new(x).to_ec() for x in scalars])
        st = (EcExprStep * len(steps))(*[EcExprStep(*q) for q in steps])
        mn, mx = EcValue(), EcValue()
        check(lib().ec_sharded_expr_min_max(self.handle, dt, p, m, k, sc, len(scalars), st, len(steps), (C.c_size_t * self.n)(*lens),
                                            C.byref(mn), C.byref(mx)))
        return B.CellValue.from_ec(mn), B.CellValue.from_ec(mx)

    def program_host(self, arrays, scalars, steps, rows: int, cols: int, nodata=None, out_nodata=None, want_mask=False, chunk_cells: int = 0):
        """`fused.program_host` / `program_host_masked` over all the GPUs of the group: the row-blocks of the host arrays
        (rows x cols cells each) stream through their own device's PCIe link side by side (`ec_sharded_host_expr`)."""
        import numpy as np
        from ._ffi import EcExprStep
        arrays = [np.ascontiguousarray(a).reshape(-1) for a in arrays]
        n = rows * cols
        assert all(a.size >= n for a in arrays)
        k = len(arrays)
        cts = [B.cell_type_of(a.dtype) for a in arrays]
        dt = (C.c_uint8 * k)(*cts)
        p = (C.c_void_p * k)(*[a.ctypes.data for a in arrays])
        nd = None
        if nodata is not None:
            nds = [None if v is None else B.CellValue(ct, v).to_ec() for ct, v in zip(cts, nodata)]
            nd = (C.POINTER(EcValue) * k)(*[C.pointer(v) if v is not None else C.POINTER(EcValue)() for v in nds])
        sc = (EcValue * max(1, len(scalars)))(*[B.CellValue.new(x).to_ec() for x in scalars])
        st = (EcExprStep * len(steps))(*[EcExprStep(*q) for q in steps])
        out = np.empty(n, dtype=np.float64)
        mask = np.empty(n, dtype=np.uint8) if want_mask else None
        ond = C.c_double(out_nodata) if out_nodata is not None else None
        check(lib().ec_sharded_host_expr(self.handle, dt, p, nd, k, sc, len(scalars), st, len(steps), rows, cols, out.ctypes.data,
                                         C.byref(ond) if ond is not None else None, mask.ctypes.data if mask is not None else None, chunk_cells))
        return (out, mask.astype(bool)) if want_mask else out

    def min_max(self, sb: ShardedBuffer, mask: ShardedBuffer = None):
        mn, mx = EcValue(), EcValue()
        check(lib().ec_sharded_min_max(self.handle, sb.ct, sb.ptrs, mask.ptrs if mask is not None else None,
                                       (C.c_size_t * self.n)(*sb.lens), C.byref(mn), C.byref(mx)))
        return B.CellValue.from_ec(mn), B.CellValue.from_ec(mx)

    def counts(self, mask: ShardedBuffer) -> tuple[int, int]:
        t, f = C.c_uint64(), C.c_uint64()
        check(lib().ec_sharded_counts(self.handle, mask.ptrs, (C.c_size_t * self.n)(*mask.lens), C.byref(t), C.byref(f)))
        return t.value, f.value
