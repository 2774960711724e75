//! Raw `extern "C"` declarations of include/erased_cells.h (ABI version 1).
//! Every data pointer is a DEVICE pointer unless the name says `host`.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub type ec_status = i32;
pub type ec_dtype = u8; // `CellType as u8` (src/ctype.rs:16)
pub type ec_op = i32;
pub type ec_stream = *mut c_void; // hipStream_t
pub type ec_comm = *mut c_void; // ncclComm_t

/// Mirrors `ec_comm_uid` (ncclUniqueId): 128 opaque bytes rank 0 hands to the other ranks.
#[repr(C)]
#[derive(Copy, Clone)]
pub struct ec_comm_uid {
    pub bytes: [c_char; 128],
}

/// Opaque `ec_shard_group`: one process driving n GPUs.
#[repr(C)]
pub struct ec_shard_group {
    _private: [u8; 0],
}
pub const EC_GROUP_RCCL: u32 = 0;
pub const EC_GROUP_HOST_COMBINE: u32 = 1;
/// Every sharded call waits until all launch threads have issued (the form of rounds 1-2; default: fire-and-forget).
pub const EC_GROUP_BLOCKING_ISSUE: u32 = 2;
/// `ec_shard_fn`: called once per shard on that shard's launch thread.
pub type ec_shard_fn = extern "C" fn(shard: i32, device: i32, stream: ec_stream, user: *mut c_void) -> ec_status;

pub const EC_OK: ec_status = 0;
pub const EC_ERR_NARROWING: ec_status = 1;
pub const EC_ADD: ec_op = 0;
pub const EC_SUB: ec_op = 1;
pub const EC_MUL: ec_op = 2;
pub const EC_DIV: ec_op = 3;

/// Mirrors `ec_value`: tag + 8-byte payload, 16 bytes.
#[repr(C)]
#[derive(Copy, Clone)]
pub struct ec_value {
    pub dtype: u8,
    pub pad_: [u8; 7],
    pub bits: u64, // the C union; read/written through to_bits()/from_bits() of the primitive
}

/// One step of an expression program (`ec_expr`): `reg[dst] = a op b`; `a`, `b` are operand references
/// (`ec_expr_stream(k)`, `ec_expr_reg(k)`, `ec_expr_scalar(k)`).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ec_expr_step {
    pub op: i8,
    pub a: i8,
    pub b: i8,
    pub dst: i8,
}
pub const fn ec_expr_stream(k: i8) -> i8 {
    k
}
pub const fn ec_expr_reg(k: i8) -> i8 {
    4 + k
}
pub const fn ec_expr_scalar(k: i8) -> i8 {
    8 + k
}
pub const EC_EXPR_MAX_STREAMS: usize = 4;
pub const EC_EXPR_REGS: usize = 4;
pub const EC_EXPR_MAX_SCALARS: usize = 8;
pub const EC_EXPR_MAX_STEPS: usize = 16;

extern "C" {
    pub fn ec_abi_version() -> i32;
    pub fn ec_init(device: i32) -> ec_status;
    pub fn ec_set_device(device: i32) -> ec_status;
    pub fn ec_get_device(device: *mut i32) -> ec_status;
    pub fn ec_shutdown() -> ec_status;
    pub fn ec_last_error_string() -> *const c_char;
    pub fn ec_last_narrowing(src: *mut ec_dtype, dst: *mut ec_dtype) -> ec_status;
    pub fn ec_device_info(n_cu: *mut i32, hbm_bytes: *mut u64, name: *mut c_char, name_cap: usize) -> ec_status;

    pub fn ec_alloc(dptr: *mut *mut c_void, bytes: usize) -> ec_status;
    pub fn ec_free(dptr: *mut c_void) -> ec_status;
    pub fn ec_alloc_async(dptr: *mut *mut c_void, bytes: usize, s: ec_stream) -> ec_status;
    pub fn ec_free_async(dptr: *mut c_void, s: ec_stream) -> ec_status;
    pub fn ec_free_ordered(dptr: *mut c_void, alloc_stream: ec_stream, last_use_stream: ec_stream) -> ec_status;
    pub fn ec_pool_trim(keep_bytes: usize) -> ec_status;
    pub fn ec_stream_create(out: *mut ec_stream) -> ec_status;
    pub fn ec_prepare_stream(s: ec_stream) -> ec_status;
    pub fn ec_release_stream(s: ec_stream) -> ec_status;
    pub fn ec_stream_destroy(s: ec_stream) -> ec_status;
    pub fn ec_stream_sync(s: ec_stream) -> ec_status;
    pub fn ec_upload(dst_dev: *mut c_void, src_host: *const c_void, bytes: usize, s: ec_stream) -> ec_status;
    pub fn ec_download(dst_host: *mut c_void, src_dev: *const c_void, bytes: usize, s: ec_stream) -> ec_status;
    pub fn ec_copy(dst_dev: *mut c_void, src_dev: *const c_void, bytes: usize, s: ec_stream) -> ec_status;

    pub fn ec_union(a: ec_dtype, b: ec_dtype) -> ec_dtype;
    pub fn ec_can_fit_into(src: ec_dtype, dst: ec_dtype) -> i32;
    pub fn ec_size_of(t: ec_dtype) -> usize;
    pub fn ec_neg_result_type(t: ec_dtype) -> ec_dtype;
    pub fn ec_min_value(t: ec_dtype, out: *mut ec_value) -> ec_status;
    pub fn ec_max_value(t: ec_dtype, out: *mut ec_value) -> ec_status;
    pub fn ec_nodata_default(t: ec_dtype, out: *mut ec_value) -> ec_status;
    pub fn ec_value_convert(v: *const ec_value, dst: ec_dtype, out: *mut ec_value) -> ec_status;
    pub fn ec_value_to_f64(v: *const ec_value) -> f64;

    pub fn ec_binop(op: ec_op, lt: ec_dtype, l: *const c_void, rt: ec_dtype, r: *const c_void, n: usize,
                    out: *mut f64, s: ec_stream) -> ec_status;
    pub fn ec_binop_scalar(op: ec_op, lt: ec_dtype, l: *const c_void, n: usize, rhs: *const ec_value,
                           out: *mut f64, s: ec_stream) -> ec_status;
    pub fn ec_masked_binop(op: ec_op, lt: ec_dtype, l: *const c_void, lmask: *const u8, rt: ec_dtype,
                           r: *const c_void, rmask: *const u8, n: usize, out: *mut f64, out_mask: *mut u8,
                           s: ec_stream) -> ec_status;
    pub fn ec_neg(t: ec_dtype, input: *const c_void, n: usize, out: *mut c_void, s: ec_stream) -> ec_status;
    pub fn ec_convert(st: ec_dtype, src: *const c_void, dt: ec_dtype, dst: *mut c_void, n: usize, s: ec_stream) -> ec_status;
    pub fn ec_fill(t: ec_dtype, dst: *mut c_void, n: usize, value: *const ec_value, s: ec_stream) -> ec_status;

    pub fn ec_min_max(t: ec_dtype, p: *const c_void, mask_or_null: *const u8, n: usize, mn: *mut ec_value,
                      mx: *mut ec_value, s: ec_stream) -> ec_status;
    pub fn ec_min_max_keys(t: ec_dtype, p: *const c_void, mask_or_null: *const u8, n: usize, keys2_dev: *mut i64,
                           s: ec_stream) -> ec_status;
    pub fn ec_min_max_decode(t: ec_dtype, keys2_host: *const i64, mn: *mut ec_value, mx: *mut ec_value) -> ec_status;

    pub fn ec_mask_from_nodata(t: ec_dtype, p: *const c_void, n: usize, nd_or_null: *const ec_value, mask: *mut u8,
                               s: ec_stream) -> ec_status;
    pub fn ec_mask_select(t: ec_dtype, p: *const c_void, mask: *const u8, n: usize, nd_or_null: *const ec_value,
                          out: *mut c_void, s: ec_stream) -> ec_status;
    pub fn ec_mask_and(l: *const u8, r: *const u8, n: usize, out: *mut u8, s: ec_stream) -> ec_status;
    pub fn ec_mask_or(l: *const u8, r: *const u8, n: usize, out: *mut u8, s: ec_stream) -> ec_status;
    pub fn ec_mask_not(m: *const u8, n: usize, out: *mut u8, s: ec_stream) -> ec_status;
    pub fn ec_mask_counts(m: *const u8, n: usize, n_true: *mut u64, n_false: *mut u64, s: ec_stream) -> ec_status;
    pub fn ec_mask_counts_device(m: *const u8, n: usize, counts2_dev: *mut u64, s: ec_stream) -> ec_status;
    pub fn ec_first_difference(t: ec_dtype, l: *const c_void, r: *const c_void, n: usize, index: *mut u64,
                               s: ec_stream) -> ec_status;
    pub fn ec_buffer_cmp(lt: ec_dtype, l: *const c_void, nl: usize, rt: ec_dtype, r: *const c_void, nr: usize,
                         ordering: *mut i32, s: ec_stream) -> ec_status;
    pub fn ec_fused(o1: ec_op, o2: ec_op, o3: ec_op, dt: *const ec_dtype, p: *const *const c_void,
                    scalars_or_null: *const ec_value, n: usize, out: *mut f64, s: ec_stream) -> ec_status;
    pub fn ec_masked_fused(o1: ec_op, o2: ec_op, o3: ec_op, dt: *const ec_dtype, p: *const *const c_void,
                           masks: *const *const u8, scalars_or_null: *const ec_value, n: usize, out: *mut f64,
                           out_mask: *mut u8, s: ec_stream) -> ec_status;
    pub fn ec_expr(dt: *const ec_dtype, p: *const *const c_void, n_streams: i32, scalars: *const ec_value, n_scalars: i32,
                   steps: *const ec_expr_step, n_steps: i32, n: usize, out: *mut f64, s: ec_stream) -> ec_status;
    pub fn ec_masked_expr(dt: *const ec_dtype, p: *const *const c_void, masks: *const *const u8, n_streams: i32,
                          scalars: *const ec_value, n_scalars: i32, steps: *const ec_expr_step, n_steps: i32, n: usize,
                          out: *mut f64, out_mask: *mut u8, s: ec_stream) -> ec_status;
    pub fn ec_expr_min_max(dt: *const ec_dtype, p: *const *const c_void, masks_or_null: *const *const u8, n_streams: i32,
                           scalars: *const ec_value, n_scalars: i32, steps: *const ec_expr_step, n_steps: i32, n: usize,
                           mn: *mut ec_value, mx: *mut ec_value, s: ec_stream) -> ec_status;
    pub fn ec_expr_min_max_keys(dt: *const ec_dtype, p: *const *const c_void, masks_or_null: *const *const u8, n_streams: i32,
                                scalars: *const ec_value, n_scalars: i32, steps: *const ec_expr_step, n_steps: i32, n: usize,
                                keys2_dev: *mut i64, s: ec_stream) -> ec_status;
    pub fn ec_sharded_expr_min_max(g: *mut ec_shard_group, dt: *const ec_dtype, p: *const *const *const c_void,
                                   masks_or_null: *const *const *const u8, n_streams: i32, scalars: *const ec_value, n_scalars: i32,
                                   steps: *const ec_expr_step, n_steps: i32, n: *const usize, mn: *mut ec_value, mx: *mut ec_value) -> ec_status;
    pub fn ec_expr_source(dt: *const ec_dtype, n_streams: i32, n_scalars: i32, steps: *const ec_expr_step, n_steps: i32,
                          arch_or_null: *const c_char, buf: *mut c_char, cap: usize, len: *mut usize) -> ec_status;
    pub fn ec_host_alloc(hptr: *mut *mut c_void, bytes: usize) -> ec_status;
    pub fn ec_host_free(hptr: *mut c_void) -> ec_status;
    pub fn ec_host_expr(dt: *const ec_dtype, p_host: *const *const c_void, n_streams: i32, scalars: *const ec_value, n_scalars: i32,
                        steps: *const ec_expr_step, n_steps: i32, n: usize, out_host: *mut f64, chunk_cells: usize) -> ec_status;
    pub fn ec_host_masked_expr(dt: *const ec_dtype, p_host: *const *const c_void, nodata: *const *const ec_value, n_streams: i32,
                               scalars: *const ec_value, n_scalars: i32, steps: *const ec_expr_step, n_steps: i32, n: usize,
                               out_host: *mut f64, out_nodata_or_null: *const f64, out_mask_host_or_null: *mut u8, chunk_cells: usize) -> ec_status;
    pub fn ec_comm_get_unique_id(uid: *mut ec_comm_uid) -> ec_status;
    pub fn ec_comm_init_rank(uid: *const ec_comm_uid, n_ranks: i32, rank: i32, comm: *mut ec_comm) -> ec_status;
    pub fn ec_comm_init_all(devices: *const i32, n: i32, comms: *mut ec_comm) -> ec_status;
    pub fn ec_comm_destroy(comm: ec_comm) -> ec_status;
    pub fn ec_allreduce_min_max_keys(comm: ec_comm, keys2_dev: *mut i64, s: ec_stream) -> ec_status;
    pub fn ec_allreduce_counts(comm: ec_comm, counts2_dev: *mut u64, s: ec_stream) -> ec_status;

    pub fn ec_shard_group_create(devices: *const i32, n: i32, flags: u32, out: *mut *mut ec_shard_group) -> ec_status;
    pub fn ec_shard_group_destroy(g: *mut ec_shard_group) -> ec_status;
    pub fn ec_shard_group_size(g: *const ec_shard_group) -> i32;
    pub fn ec_shard_group_shard(g: *const ec_shard_group, shard: i32, device: *mut i32, stream: *mut ec_stream) -> ec_status;
    pub fn ec_shard_group_foreach(g: *mut ec_shard_group, f: ec_shard_fn, user: *mut c_void) -> ec_status;
    pub fn ec_shard_group_sync(g: *mut ec_shard_group) -> ec_status;
    pub fn ec_shard_group_stat(g: *const ec_shard_group, key: *const c_char, value: *mut i64) -> ec_status;
    pub fn ec_sharded_alloc(g: *mut ec_shard_group, bytes: *const usize, dptrs: *mut *mut c_void) -> ec_status;
    pub fn ec_sharded_free(g: *mut ec_shard_group, dptrs: *const *mut c_void) -> ec_status;
    pub fn ec_sharded_upload(g: *mut ec_shard_group, dst_dev: *const *mut c_void, src_host: *const c_void,
                             byte_offsets: *const usize, bytes: *const usize) -> ec_status;
    pub fn ec_sharded_download(g: *mut ec_shard_group, dst_host: *mut c_void, src_dev: *const *const c_void,
                               byte_offsets: *const usize, bytes: *const usize) -> ec_status;
    pub fn ec_sharded_binop(g: *mut ec_shard_group, op: ec_op, lt: ec_dtype, l: *const *const c_void, rt: ec_dtype,
                            r: *const *const c_void, n: *const usize, out: *const *mut f64) -> ec_status;
    pub fn ec_sharded_masked_binop(g: *mut ec_shard_group, op: ec_op, lt: ec_dtype, l: *const *const c_void,
                                   lmask: *const *const u8, rt: ec_dtype, r: *const *const c_void, rmask: *const *const u8,
                                   n: *const usize, out: *const *mut f64, out_mask: *const *mut u8) -> ec_status;
    pub fn ec_sharded_convert(g: *mut ec_shard_group, st: ec_dtype, src: *const *const c_void, dt: ec_dtype,
                              dst: *const *mut c_void, n: *const usize) -> ec_status;
    pub fn ec_sharded_mask_from_nodata(g: *mut ec_shard_group, t: ec_dtype, p: *const *const c_void, n: *const usize,
                                       nd_or_null: *const ec_value, mask: *const *mut u8) -> ec_status;
    pub fn ec_sharded_fused(g: *mut ec_shard_group, o1: ec_op, o2: ec_op, o3: ec_op, dt: *const ec_dtype,
                            p: *const *const *const c_void, masks_or_null: *const *const *const u8,
                            scalars_or_null: *const ec_value, n: *const usize, out: *const *mut f64,
                            out_mask_or_null: *const *mut u8) -> ec_status;
    pub fn ec_sharded_expr(g: *mut ec_shard_group, dt: *const ec_dtype, p: *const *const *const c_void,
                           masks_or_null: *const *const *const u8, n_streams: i32, scalars: *const ec_value, n_scalars: i32,
                           steps: *const ec_expr_step, n_steps: i32, n: *const usize, out: *const *mut f64,
                           out_mask_or_null: *const *mut u8) -> ec_status;
    pub fn ec_sharded_host_expr(g: *mut ec_shard_group, dt: *const ec_dtype, p_host: *const *const c_void, nodata_or_null: *const *const ec_value,
                                n_streams: i32, scalars: *const ec_value, n_scalars: i32, steps: *const ec_expr_step, n_steps: i32,
                                n_rows: u64, n_cols: u64, out_host: *mut f64, out_nodata_or_null: *const f64,
                                out_mask_host_or_null: *mut u8, chunk_cells: usize) -> ec_status;
    pub fn ec_sharded_min_max(g: *mut ec_shard_group, t: ec_dtype, p: *const *const c_void,
                              masks_or_null: *const *const u8, n: *const usize, mn: *mut ec_value,
                              mx: *mut ec_value) -> ec_status;
    pub fn ec_sharded_counts(g: *mut ec_shard_group, masks: *const *const u8, n: *const usize, n_true: *mut u64,
                             n_false: *mut u64) -> ec_status;
    pub fn ec_shard_range(n_rows: u64, n_cols: u64, shard: u32, n_shards: u32, cell_offset: *mut u64,
                          cell_len: *mut u64) -> ec_status;

    // test support: deterministic device-side inputs and tuning knobs
    pub fn ec_synth_fill(t: ec_dtype, dst: *mut c_void, n: usize, seed: u64, base: u64, lo: f64, hi: f64,
                         s: ec_stream) -> ec_status;
    pub fn ec_synth_mask(dst: *mut u8, n: usize, seed: u64, base: u64, pct_nodata: u32, s: ec_stream) -> ec_status;
    pub fn ec_tune_set(key: *const c_char, value: i64) -> ec_status;
    pub fn ec_stat_get(key: *const c_char, value: *mut i64) -> ec_status;
}
