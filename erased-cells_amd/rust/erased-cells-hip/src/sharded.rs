//! Row-block sharding across the GPUs of one node, one process per GPU.
//! Element-wise work is local to a shard; only the two scalar statistics cross GPUs.
use crate::ffi::*;
use crate::{must, stream, CellBuffer, CellValue, Mask};
use std::os::raw::c_void;

/// Contiguous cell range `(offset, len)` of shard `shard` of `n_shards` for an `n_rows x n_cols` raster.
pub fn shard_range(n_rows: u64, n_cols: u64, shard: u32, n_shards: u32) -> (u64, u64) {
    let (mut off, mut len) = (0u64, 0u64);
    must(unsafe { ec_shard_range(n_rows, n_cols, shard, n_shards, &mut off, &mut len) }, "ec_shard_range");
    (off, len)
}

/// Global `min_max` of a row-sharded buffer: local order keys, all-reduce MAX of two int64 words over
/// xGMI on the caller's `ncclComm_t`, decode.  `keys2_dev` is a 16-byte device scratch owned by the caller.
pub fn min_max(local: &CellBuffer, mask: Option<&Mask>, rccl_comm: *mut c_void, keys2_dev: *mut i64) -> (CellValue, CellValue) {
    let m = mask.map_or(std::ptr::null(), |m| m.dev_ptr());
    must(unsafe { ec_min_max_keys(local.ct as u8, local.dev_ptr(), m, local.len(), keys2_dev, stream()) }, "ec_min_max_keys");
    must(unsafe { ec_allreduce_min_max_keys(rccl_comm, keys2_dev, stream()) }, "ec_allreduce_min_max_keys");
    let mut host = [0i64; 2];
    must(unsafe { ec_download(host.as_mut_ptr() as *mut c_void, keys2_dev as *const c_void, 16, stream()) }, "ec_download");
    let (mut mn, mut mx) = (CellValue::UInt8(0).to_ffi(), CellValue::UInt8(0).to_ffi());
    must(unsafe { ec_min_max_decode(local.ct as u8, host.as_ptr(), &mut mn, &mut mx) }, "ec_min_max_decode");
    (CellValue::from_ffi(&mn), CellValue::from_ffi(&mx))
}

/// Global `(true, false)` counts of a row-sharded mask: all-reduce SUM of two uint64 words.
pub fn counts(local: &Mask, rccl_comm: *mut c_void, counts2_dev: *mut u64) -> (u64, u64) {
    must(unsafe { ec_mask_counts_device(local.dev_ptr(), local.len(), counts2_dev, stream()) }, "ec_mask_counts_device");
    must(unsafe { ec_allreduce_counts(rccl_comm, counts2_dev, stream()) }, "ec_allreduce_counts");
    let mut host = [0u64; 2];
    must(unsafe { ec_download(host.as_mut_ptr() as *mut c_void, counts2_dev as *const c_void, 16, stream()) }, "ec_download");
    (host[0], host[1])
}
