//! Row-block sharding across the GPUs of one node.
//!
//! Element-wise work is local to a shard; only the two scalar statistics cross GPUs — `min_max` as an
//! all-reduce(MAX) of two order-preserving int64 keys, `counts` as an all-reduce(SUM) of two uint64, both over
//! xGMI through RCCL.  Two shapes:
//!
//! * one process per GPU — [`Communicator`]: rank 0 makes a [`UniqueId`], the host program hands its 128 bytes to
//!   the other ranks, every rank joins; then [`Communicator::min_max`] / [`Communicator::counts`] on its own shard;
//! * one process driving all GPUs — [`ShardGroup`]: per device a launch thread, a stream and a communicator of
//!   the clique, owned by the library; [`ShardGroup::scatter`] cuts a host `Vec` into row-blocks.
//!
//! Neither takes a foreign `ncclComm_t`: the library builds its communicators itself.
use crate::device::stream;
use crate::error::{check, must, Result};
use crate::ffi::*;
use crate::{BufferOps, CellBuffer, CellEncoding, CellType, CellValue};
#[cfg(feature = "masked")]
use crate::Mask;
use std::os::raw::c_void;
use std::ptr;

/// Contiguous cell range `(offset, len)` of shard `shard` of `n_shards` for an `n_rows x n_cols` raster: rows
/// `[g*R/G, (g+1)*R/G)`, the first `R mod G` shards one row longer.
pub fn shard_range(n_rows: u64, n_cols: u64, shard: u32, n_shards: u32) -> (u64, u64) {
    let (mut off, mut len) = (0u64, 0u64);
    must(unsafe { ec_shard_range(n_rows, n_cols, shard, n_shards, &mut off, &mut len) }, "ec_shard_range");
    (off, len)
}

/// The 128 bytes rank 0 hands to the other ranks (a file, a socket, an environment variable ...).
#[derive(Copy, Clone)]
pub struct UniqueId(pub [u8; 128]);

impl UniqueId {
    /// Call on rank 0 only.
    pub fn new() -> Result<Self> {
        let mut uid = ec_comm_uid { bytes: [0; 128] };
        check(unsafe { ec_comm_get_unique_id(&mut uid) })?;
        Ok(UniqueId(uid.bytes.map(|b| b as u8)))
    }
}

/// This rank's end of an RCCL communicator (one process per GPU), with its 32-byte device payload slot.
pub struct Communicator {
    comm: ec_comm,
    payload: *mut c_void,
}

impl Communicator {
    /// Blocks until all `n_ranks` ranks have joined; the calling thread's device ([`crate::init`]) is this rank's GPU.
    pub fn join(uid: &UniqueId, n_ranks: i32, rank: i32) -> Result<Self> {
        let raw = ec_comm_uid { bytes: uid.0.map(|b| b as std::os::raw::c_char) };
        let mut comm: ec_comm = ptr::null_mut();
        check(unsafe { ec_comm_init_rank(&raw, n_ranks, rank, &mut comm) })?;
        let mut payload = ptr::null_mut();
        check(unsafe { ec_alloc(&mut payload, 32) })?;
        Ok(Self { comm, payload })
    }

    /// Global `min_max` of the raster whose local row-block is `local` (collective: every rank calls it).
    #[cfg(feature = "masked")]
    pub fn min_max(&self, local: &CellBuffer, mask: Option<&Mask>) -> (CellValue, CellValue) {
        self.min_max_raw(local, mask.map_or(ptr::null(), |m| m.dev_ptr()))
    }
    #[cfg(not(feature = "masked"))]
    pub fn min_max(&self, local: &CellBuffer) -> (CellValue, CellValue) {
        self.min_max_raw(local, ptr::null())
    }
    fn min_max_raw(&self, local: &CellBuffer, mask: *const u8) -> (CellValue, CellValue) {
        let keys = self.payload as *mut i64;
        must(unsafe { ec_min_max_keys(local.ct as u8, local.dev_ptr(), mask, local.len(), keys, stream()) }, "ec_min_max_keys");
        must(unsafe { ec_allreduce_min_max_keys(self.comm, keys, stream()) }, "ec_allreduce_min_max_keys");
        let mut host = [0i64; 2];
        must(unsafe { ec_download(host.as_mut_ptr() as *mut c_void, keys as *const c_void, 16, stream()) }, "ec_download");
        decode_keys(local.ct, &host)
    }

    /// Global `(data, nodata)` counts of a row-sharded mask (collective).
    #[cfg(feature = "masked")]
    pub fn counts(&self, local: &Mask) -> (u64, u64) {
        let c = self.payload as *mut u64;
        must(unsafe { ec_mask_counts_device(local.dev_ptr(), local.len(), c, stream()) }, "ec_mask_counts_device");
        must(unsafe { ec_allreduce_counts(self.comm, c, stream()) }, "ec_allreduce_counts");
        let mut host = [0u64; 2];
        must(unsafe { ec_download(host.as_mut_ptr() as *mut c_void, c as *const c_void, 16, stream()) }, "ec_download");
        (host[0], host[1])
    }
}

impl Drop for Communicator {
    fn drop(&mut self) {
        unsafe {
            ec_free(self.payload);
            ec_comm_destroy(self.comm);
        }
    }
}

fn decode_keys(ct: CellType, keys: &[i64; 2]) -> (CellValue, CellValue) {
    let (mut mn, mut mx) = (CellValue::UInt8(0).to_ffi(), CellValue::UInt8(0).to_ffi());
    must(unsafe { ec_min_max_decode(ct as u8, keys.as_ptr(), &mut mn, &mut mx) }, "ec_min_max_decode");
    (CellValue::from_ffi(&mn), CellValue::from_ffi(&mx))
}

/// One process driving `n` GPUs: shard `i` of every sharded buffer lives on device `i` of the group.
pub struct ShardGroup {
    g: *mut ec_shard_group,
    n: usize,
}

/// A raster cut into row-blocks, one device block per GPU of a [`ShardGroup`].
pub struct ShardedCellBuffer<'g> {
    group: &'g ShardGroup,
    ct: CellType,
    ptrs: Vec<*mut c_void>,
    lens: Vec<usize>,
}

impl ShardGroup {
    /// `devices` are HIP device indices; RCCL needs them distinct.  Initialises each device.
    pub fn new(devices: &[i32]) -> Result<Self> {
        let mut g = ptr::null_mut();
        check(unsafe { ec_shard_group_create(devices.as_ptr(), devices.len() as i32, EC_GROUP_RCCL, &mut g) })?;
        Ok(Self { g, n: devices.len() })
    }

    pub fn len(&self) -> usize {
        self.n
    }

    pub fn is_empty(&self) -> bool {
        self.n == 0
    }

    /// Cut `data` (row-major, `n_rows x n_cols`) into contiguous row-blocks and upload block `i` to device `i`.
    pub fn scatter<T: CellEncoding>(&self, data: &[T], n_rows: u64, n_cols: u64) -> Result<ShardedCellBuffer<'_>> {
        assert_eq!(data.len() as u64, n_rows * n_cols);
        let sz = std::mem::size_of::<T>();
        let ranges: Vec<(u64, u64)> = (0..self.n).map(|i| shard_range(n_rows, n_cols, i as u32, self.n as u32)).collect();
        let lens: Vec<usize> = ranges.iter().map(|r| r.1 as usize).collect();
        let bytes: Vec<usize> = lens.iter().map(|l| l * sz).collect();
        let offs: Vec<usize> = ranges.iter().map(|r| r.0 as usize * sz).collect();
        let mut ptrs: Vec<*mut c_void> = vec![ptr::null_mut(); self.n];
        check(unsafe { ec_sharded_alloc(self.g, bytes.as_ptr(), ptrs.as_mut_ptr()) })?;
        check(unsafe { ec_sharded_upload(self.g, ptrs.as_ptr(), data.as_ptr() as *const c_void, offs.as_ptr(), bytes.as_ptr()) })?;
        Ok(ShardedCellBuffer { group: self, ct: T::cell_type(), ptrs, lens })
    }

    /// Wait until everything queued so far has been issued and every shard's stream has drained.  The element-wise
    /// sharded calls are fire-and-forget: a failure inside one of them (a launch error) is reported here, once.
    pub fn sync(&self) -> Result<()> {
        check(unsafe { ec_shard_group_sync(self.g) })
    }
}

impl Drop for ShardGroup {
    fn drop(&mut self) {
        unsafe { ec_shard_group_destroy(self.g) };
    }
}

impl ShardedCellBuffer<'_> {
    pub fn cell_type(&self) -> CellType {
        self.ct
    }

    pub fn len(&self) -> usize {
        self.lens.iter().sum()
    }

    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }

    fn const_ptrs(&self) -> Vec<*const c_void> {
        self.ptrs.iter().map(|p| *p as *const c_void).collect()
    }

    fn alloc_like(&self, ct: CellType) -> Result<Self> {
        let bytes: Vec<usize> = self.lens.iter().map(|l| l * ct.size_of()).collect();
        let mut ptrs: Vec<*mut c_void> = vec![ptr::null_mut(); self.lens.len()];
        check(unsafe { ec_sharded_alloc(self.group.g, bytes.as_ptr(), ptrs.as_mut_ptr()) })?;
        Ok(Self { group: self.group, ct, ptrs, lens: self.lens.clone() })
    }

    /// `self op rhs` on every shard (no communication); both rasters must be cut the same way.
    pub fn binop(&self, op: ec_op, rhs: &Self) -> Result<Self> {
        assert_eq!(self.lens, rhs.lens, "operands must be sharded identically");
        let out = self.alloc_like(CellType::Float64)?;
        let outs: Vec<*mut f64> = out.ptrs.iter().map(|p| *p as *mut f64).collect();
        check(unsafe {
            ec_sharded_binop(self.group.g, op, self.ct as u8, self.const_ptrs().as_ptr(), rhs.ct as u8, rhs.const_ptrs().as_ptr(),
                             self.lens.as_ptr(), outs.as_ptr())
        })?;
        Ok(out)
    }

    /// `BufferOps::min_max` of the whole raster: per-shard keys, one all-reduce over xGMI, decode.
    pub fn min_max(&self) -> Result<(CellValue, CellValue)> {
        let (mut mn, mut mx) = (CellValue::UInt8(0).to_ffi(), CellValue::UInt8(0).to_ffi());
        check(unsafe {
            ec_sharded_min_max(self.group.g, self.ct as u8, self.const_ptrs().as_ptr(), ptr::null(), self.lens.as_ptr(), &mut mn, &mut mx)
        })?;
        Ok((CellValue::from_ffi(&mn), CellValue::from_ffi(&mx)))
    }

    /// Gather the shards back into one host `Vec` (the type must be the buffer's own cell type).
    pub fn to_vec<T: CellEncoding + Default>(&self) -> Result<Vec<T>> {
        assert_eq!(self.ct, T::cell_type());
        let sz = std::mem::size_of::<T>();
        let mut out = vec![T::default(); self.len()];
        let bytes: Vec<usize> = self.lens.iter().map(|l| l * sz).collect();
        let mut offs = Vec::with_capacity(self.lens.len());
        let mut acc = 0usize;
        for l in &self.lens {
            offs.push(acc * sz);
            acc += l;
        }
        check(unsafe {
            ec_sharded_download(self.group.g, out.as_mut_ptr() as *mut c_void, self.const_ptrs().as_ptr(), offs.as_ptr(), bytes.as_ptr())
        })?;
        Ok(out)
    }
}

impl Drop for ShardedCellBuffer<'_> {
    fn drop(&mut self) {
        unsafe {
            ec_shard_group_sync(self.group.g);
            ec_sharded_free(self.group.g, self.ptrs.as_ptr());
        }
    }
}

/// A single-device view, for code that mixes the two shapes: shard `i` of a sharded buffer as a borrowed
/// `(device pointer, len)` pair on device `i`'s stream.
pub fn shard_of<'a>(b: &'a ShardedCellBuffer<'_>, i: usize) -> (*const c_void, usize) {
    (b.ptrs[i] as *const c_void, b.lens[i])
}
