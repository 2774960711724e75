"""ctypes binding of liberased_cells_hip.so (include/erased_cells.h).

There is no fallback: if the shared library is missing or a symbol is absent the
import fails loudly, and every compute call raises unless a HIP device is bound.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
AMD_DIR = os.path.dirname(os.path.dirname(_PKG))          # erased-cells_amd/
REPO = os.path.dirname(AMD_DIR)
SO_PATH = os.environ.get("EC_HIP_LIB") or os.path.join(AMD_DIR, "liberased_cells_hip.so")  # EC_HIP_LIB: A/B builds
CSRC = os.path.join(AMD_DIR, "csrc")

EC_OK, EC_ERR_NARROWING, EC_ERR_UNSUPPORTED_TYPE, EC_ERR_LENGTH, EC_ERR_HIP, EC_ERR_RCCL, EC_ERR_ARG, \
    EC_ERR_NOT_INITIALIZED = range(8)


class _Payload(C.Union):
    _fields_ = [("u8", C.c_uint8), ("u16", C.c_uint16), ("u32", C.c_uint32), ("u64", C.c_uint64),
                ("i8", C.c_int8), ("i16", C.c_int16), ("i32", C.c_int32), ("i64", C.c_int64),
                ("f32", C.c_float), ("f64", C.c_double), ("bits", C.c_uint64)]


class EcValue(C.Structure):
    """ec_value: 16-byte tagged scalar (CellValue, src/value.rs:12-20)."""
    _fields_ = [("dtype", C.c_uint8), ("pad_", C.c_uint8 * 7), ("v", _Payload)]


class EcExprStep(C.Structure):
    """ec_expr_step: reg[dst] = a op b (include/erased_cells.h)."""
    _fields_ = [("op", C.c_int8), ("a", C.c_int8), ("b", C.c_int8), ("dst", C.c_int8)]


VP, SZ, I32, U8P = C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p
PV = C.POINTER(EcValue)
PVP, PSZ = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)


class EcCommUid(C.Structure):
    """ec_comm_uid (ncclUniqueId): 128 opaque bytes rank 0 hands to the other ranks."""
    _fields_ = [("bytes", C.c_char * 128)]


# ec_shard_fn: ec_status fn(int32 shard, int32 device, ec_stream stream, void* user)
SHARD_FN = C.CFUNCTYPE(C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p)

# name -> (restype, argtypes): every symbol include/erased_cells.h declares.
SIGNATURES = {
    "ec_union": (C.c_uint8, [C.c_uint8, C.c_uint8]),
    "ec_can_fit_into": (I32, [C.c_uint8, C.c_uint8]),
    "ec_size_of": (SZ, [C.c_uint8]),
    "ec_neg_result_type": (C.c_uint8, [C.c_uint8]),
    "ec_min_value": (I32, [C.c_uint8, PV]),
    "ec_max_value": (I32, [C.c_uint8, PV]),
    "ec_nodata_default": (I32, [C.c_uint8, PV]),
    "ec_value_convert": (I32, [PV, C.c_uint8, PV]),
    "ec_value_to_f64": (C.c_double, [PV]),
    "ec_shard_range": (I32, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ec_abi_version": (I32, []),
    "ec_init": (I32, [I32]),
    "ec_set_device": (I32, [I32]),
    "ec_get_device": (I32, [C.POINTER(I32)]),
    "ec_shutdown": (I32, []),
    "ec_last_error_string": (C.c_char_p, []),
    "ec_last_narrowing": (I32, [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]),
    "ec_device_info": (I32, [C.POINTER(I32), C.POINTER(C.c_uint64), C.c_char_p, SZ]),
    "ec_alloc": (I32, [C.POINTER(VP), SZ]),
    "ec_free": (I32, [VP]),
    "ec_alloc_async": (I32, [C.POINTER(VP), SZ, VP]),
    "ec_free_async": (I32, [VP, VP]),
    "ec_free_ordered": (I32, [VP, VP, VP]),
    "ec_pool_trim": (I32, [SZ]),
    "ec_upload": (I32, [VP, VP, SZ, VP]),
    "ec_download": (I32, [VP, VP, SZ, VP]),
    "ec_copy": (I32, [VP, VP, SZ, VP]),
    "ec_stream_create": (I32, [C.POINTER(VP)]),
    "ec_prepare_stream": (I32, [VP]),
    "ec_release_stream": (I32, [VP]),
    "ec_stream_destroy": (I32, [VP]),
    "ec_stream_sync": (I32, [VP]),
    "ec_binop": (I32, [I32, C.c_uint8, VP, C.c_uint8, VP, SZ, VP, VP]),
    "ec_binop_scalar": (I32, [I32, C.c_uint8, VP, SZ, PV, VP, VP]),
    "ec_masked_binop": (I32, [I32, C.c_uint8, VP, U8P, C.c_uint8, VP, U8P, SZ, VP, U8P, VP]),
    "ec_fused": (I32, [I32, I32, I32, C.POINTER(C.c_uint8), C.POINTER(VP), PV, SZ, VP, VP]),
    "ec_masked_fused": (I32, [I32, I32, I32, C.POINTER(C.c_uint8), C.POINTER(VP), C.POINTER(VP), PV, SZ, VP, U8P, VP]),
    "ec_expr": (I32, [C.POINTER(C.c_uint8), C.POINTER(VP), I32, PV, I32, C.POINTER(EcExprStep), I32, SZ, VP, VP]),
    "ec_expr_min_max": (I32, [C.POINTER(C.c_uint8), C.POINTER(VP), C.POINTER(VP), I32, PV, I32, C.POINTER(EcExprStep), I32, SZ, PV, PV, VP]),
    "ec_expr_min_max_keys": (I32, [C.POINTER(C.c_uint8), C.POINTER(VP), C.POINTER(VP), I32, PV, I32, C.POINTER(EcExprStep), I32, SZ, VP, VP]),
    "ec_sharded_expr_min_max": (I32, [VP, C.POINTER(C.c_uint8), C.POINTER(PVP), C.POINTER(PVP), I32, PV, I32, C.POINTER(EcExprStep), I32, PSZ, PV, PV]),
    "ec_expr_source": (I32, [C.POINTER(C.c_uint8), I32, I32, C.POINTER(EcExprStep), I32, C.c_char_p, C.c_char_p, SZ, C.POINTER(SZ)]),
    "ec_host_alloc": (I32, [PVP, SZ]),
    "ec_host_free": (I32, [VP]),
    "ec_host_masked_expr": (I32, [C.POINTER(C.c_uint8), C.POINTER(VP), C.POINTER(PV), I32, PV, I32, C.POINTER(EcExprStep), I32, SZ, VP,
                                  C.POINTER(C.c_double), VP, SZ]),
    "ec_host_expr": (I32, [C.POINTER(C.c_uint8), C.POINTER(VP), I32, PV, I32, C.POINTER(EcExprStep), I32, SZ, VP, SZ]),
    "ec_masked_expr": (I32, [C.POINTER(C.c_uint8), C.POINTER(VP), C.POINTER(VP), I32, PV, I32, C.POINTER(EcExprStep), I32, SZ, VP, U8P, VP]),
    "ec_neg": (I32, [C.c_uint8, VP, SZ, VP, VP]),
    "ec_convert": (I32, [C.c_uint8, VP, C.c_uint8, VP, SZ, VP]),
    "ec_fill": (I32, [C.c_uint8, VP, SZ, PV, VP]),
    "ec_min_max": (I32, [C.c_uint8, VP, U8P, SZ, PV, PV, VP]),
    "ec_min_max_keys": (I32, [C.c_uint8, VP, U8P, SZ, VP, VP]),
    "ec_min_max_decode": (I32, [C.c_uint8, C.POINTER(C.c_int64), PV, PV]),
    "ec_allreduce_min_max_keys": (I32, [VP, VP, VP]),
    "ec_allreduce_counts": (I32, [VP, VP, VP]),
    "ec_comm_get_unique_id": (I32, [C.POINTER(EcCommUid)]),
    "ec_comm_init_rank": (I32, [C.POINTER(EcCommUid), I32, I32, PVP]),
    "ec_comm_init_all": (I32, [C.POINTER(I32), I32, PVP]),
    "ec_comm_destroy": (I32, [VP]),
    "ec_shard_group_create": (I32, [C.POINTER(I32), I32, C.c_uint32, PVP]),
    "ec_shard_group_destroy": (I32, [VP]),
    "ec_shard_group_size": (I32, [VP]),
    "ec_shard_group_shard": (I32, [VP, I32, C.POINTER(I32), PVP]),
    "ec_shard_group_foreach": (I32, [VP, SHARD_FN, VP]),
    "ec_shard_group_sync": (I32, [VP]),
    "ec_shard_group_stat": (I32, [VP, C.c_char_p, C.POINTER(C.c_int64)]),
    "ec_sharded_alloc": (I32, [VP, PSZ, PVP]),
    "ec_sharded_free": (I32, [VP, PVP]),
    "ec_sharded_upload": (I32, [VP, PVP, VP, PSZ, PSZ]),
    "ec_sharded_download": (I32, [VP, VP, PVP, PSZ, PSZ]),
    "ec_sharded_binop": (I32, [VP, I32, C.c_uint8, PVP, C.c_uint8, PVP, PSZ, PVP]),
    "ec_sharded_masked_binop": (I32, [VP, I32, C.c_uint8, PVP, PVP, C.c_uint8, PVP, PVP, PSZ, PVP, PVP]),
    "ec_sharded_convert": (I32, [VP, C.c_uint8, PVP, C.c_uint8, PVP, PSZ]),
    "ec_sharded_mask_from_nodata": (I32, [VP, C.c_uint8, PVP, PSZ, PV, PVP]),
    "ec_sharded_host_expr": (I32, [VP, C.POINTER(C.c_uint8), C.POINTER(VP), C.POINTER(PV), I32, PV, I32, C.POINTER(EcExprStep), I32, C.c_uint64,
                                   C.c_uint64, VP, C.POINTER(C.c_double), VP, SZ]),
    "ec_sharded_expr": (I32, [VP, C.POINTER(C.c_uint8), C.POINTER(PVP), C.POINTER(PVP), I32, PV, I32, C.POINTER(EcExprStep), I32, PSZ, PVP, PVP]),
    "ec_sharded_fused": (I32, [VP, I32, I32, I32, C.POINTER(C.c_uint8), C.POINTER(PVP), C.POINTER(PVP), PV, PSZ, PVP, PVP]),
    "ec_sharded_min_max": (I32, [VP, C.c_uint8, PVP, PVP, PSZ, PV, PV]),
    "ec_sharded_counts": (I32, [VP, PVP, PSZ, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ec_buffer_cmp": (I32, [C.c_uint8, VP, SZ, C.c_uint8, VP, SZ, C.POINTER(I32), VP]),
    "ec_first_difference": (I32, [C.c_uint8, VP, VP, SZ, C.POINTER(C.c_uint64), VP]),
    "ec_mask_from_nodata": (I32, [C.c_uint8, VP, SZ, PV, U8P, VP]),
    "ec_mask_select": (I32, [C.c_uint8, VP, U8P, SZ, PV, VP, VP]),
    "ec_mask_and": (I32, [U8P, U8P, SZ, U8P, VP]),
    "ec_mask_or": (I32, [U8P, U8P, SZ, U8P, VP]),
    "ec_mask_not": (I32, [U8P, SZ, U8P, VP]),
    "ec_mask_counts": (I32, [U8P, SZ, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), VP]),
    "ec_mask_counts_device": (I32, [U8P, SZ, VP, VP]),
    "ec_synth_fill": (I32, [C.c_uint8, VP, SZ, C.c_uint64, C.c_uint64, C.c_double, C.c_double, VP]),
    "ec_synth_mask": (I32, [U8P, SZ, C.c_uint64, C.c_uint64, C.c_uint32, VP]),
    "ec_tune_set": (I32, [C.c_char_p, C.c_int64]),
    "ec_stat_get": (I32, [C.c_char_p, C.POINTER(C.c_int64)]),
}


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, f"-j{jobs}", "-s"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean", "-s"])
    subprocess.check_call(args)
    return SO_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and loads it by a different
        # file name than this library's DT_NEEDED entry.  If torch comes second it brings up a second
        # runtime copy that cannot see the device; if torch is loaded first the dynamic loader resolves
        # our libamdhip64.so.7 to torch's copy and both share one runtime.  So: torch first, when present.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(SO_PATH):
            raise ImportError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"(or `make -C {CSRC}`); there is no CPU fallback")
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class EcError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"[ec_status {status}] {message}")
        self.status = status


class NarrowingError(EcError):
    """Error::NarrowingError{src,dst} (src/error.rs:14-15)."""

    def __init__(self, message: str, src: int, dst: int):
        super().__init__(EC_ERR_NARROWING, message)
        self.src, self.dst = src, dst


def check(status: int) -> None:
    if status == EC_OK:
        return
    L = lib()
    msg = (L.ec_last_error_string() or b"").decode()
    if status == EC_ERR_NARROWING:
        s, d = C.c_uint8(), C.c_uint8()
        L.ec_last_narrowing(C.byref(s), C.byref(d))
        raise NarrowingError(msg, s.value, d.value)
    raise EcError(status, msg)
