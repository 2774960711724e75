//! Device-resident `CellBuffer` / `Mask` / `MaskedCellBuffer` whose operator bodies are calls into
//! liberased_cells_hip.so (C ABI: include/erased_cells.h).
//!
//! The shape follows erased-cells 0.1.1: `CellType` keeps its discriminants (they ARE the ABI dtype
//! codes), the host keeps the type tag, zip truncation, the empty-result-is-UInt8 rule and the length
//! asserts, and each per-cell loop body of the reference becomes one FFI call on HBM-resident cells.
//!
//! NOT COMPILED in the build image (no rustc there); `tests/test_abi_host.py` keeps `ffi.rs` in lockstep
//! with the header.  The same mapping, compiled and tested, exists in C++ (`host/erased_cells.hpp`) and
//! Python (`python/erased_cells_hip/buffer.py`); see INTEGRATION.md.
pub mod ffi;

mod cell_value;
mod device_buffer;
mod device_mask;
pub mod fused;
mod masked;
pub mod sharded;

pub use cell_value::CellValue;
pub use device_buffer::CellBuffer;
pub use device_mask::Mask;
pub use masked::{MaskedCellBuffer, NoData};

use ffi::*;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::ptr;

/// `CellType` (src/ctype.rs:11-20, variant order of src/lib.rs:89-98): `self as u8` is the ABI dtype code.
#[derive(Debug, Copy, Clone, PartialEq, Eq, PartialOrd, Ord, Hash)]
#[repr(u8)]
pub enum CellType { UInt8, UInt16, UInt32, UInt64, Int8, Int16, Int32, Int64, Float32, Float64 }

impl CellType {
    pub(crate) fn from_code(c: u8) -> Self {
        assert!(c <= CellType::Float64 as u8, "dtype code {c} out of range");
        unsafe { std::mem::transmute(c) }
    }
    pub fn size_of(self) -> usize { unsafe { ec_size_of(self as u8) } }
    /// src/ctype.rs:99-121
    pub fn union(self, other: Self) -> Self { Self::from_code(unsafe { ec_union(self as u8, other as u8) }) }
    /// src/ctype.rs:124-131
    pub fn can_fit_into(self, other: Self) -> bool { unsafe { ec_can_fit_into(self as u8, other as u8) != 0 } }
    /// src/ctype.rs:158-167
    pub fn min_value(self) -> CellValue {
        let mut v = CellValue::UInt8(0).to_ffi();
        unsafe { ec_min_value(self as u8, &mut v) };
        CellValue::from_ffi(&v)
    }
    /// src/ctype.rs:170-179
    pub fn max_value(self) -> CellValue {
        let mut v = CellValue::UInt8(0).to_ffi();
        unsafe { ec_max_value(self as u8, &mut v) };
        CellValue::from_ffi(&v)
    }
}

/// Implemented by the ten primitives a cell can hold (src/encoding.rs).
pub trait CellEncoding: Copy + Into<CellValue> {
    fn cell_type() -> CellType;
}
macro_rules! encoding {
    ($(($id:ident, $p:ident)),*) => { $(
        impl CellEncoding for $p { fn cell_type() -> CellType { CellType::$id } }
    )* }
}
encoding!((UInt8, u8), (UInt16, u16), (UInt32, u32), (UInt64, u64), (Int8, i8), (Int16, i16), (Int32, i32),
          (Int64, i64), (Float32, f32), (Float64, f64));

#[derive(thiserror::Error, Debug)]
pub enum Error {
    /// src/error.rs: narrowing conversions are refused (ctype lattice), never performed lossily
    #[error("Invalid narrowing from cell-type {src:?} to {dst:?}")]
    NarrowingError { src: CellType, dst: CellType },
    #[error("HIP backend: {0}")]
    Backend(String),
}
pub type Result<T> = std::result::Result<T, Error>;

pub(crate) fn check(st: ec_status) -> Result<()> {
    if st == EC_OK {
        return Ok(());
    }
    if st == EC_ERR_NARROWING {
        let (mut s, mut d) = (0u8, 0u8);
        unsafe { ec_last_narrowing(&mut s, &mut d) };
        return Err(Error::NarrowingError { src: CellType::from_code(s), dst: CellType::from_code(d) });
    }
    Err(Error::Backend(unsafe { CStr::from_ptr(ec_last_error_string()) }.to_string_lossy().into_owned()))
}

/// Arithmetic is infallible in the reference; a backend failure (no device, out of HBM) is a panic.
pub(crate) fn must(st: ec_status, what: &str) {
    if let Err(e) = check(st) {
        panic!("{what}: {e}");
    }
}

/// Bind the process to one GPU (one process per GPU; call once before any buffer exists).
pub fn init(device: i32) -> Result<()> { check(unsafe { ec_init(device) }) }

/// The stream every call of this crate is issued on (the default stream; a host that owns streams
/// passes its own `hipStream_t` through the same parameter).
pub(crate) fn stream() -> ec_stream { ptr::null_mut() }

/// One HBM allocation from the stream-ordered pool: operator results are allocated per call, as the
/// reference `collect()`s a fresh `Vec`, and hipMalloc/hipFree per operator would cost as much as the kernel.
pub(crate) struct DeviceMem {
    ptr: *mut c_void,
}
impl DeviceMem {
    pub(crate) fn new(bytes: usize) -> Self {
        let mut p = ptr::null_mut();
        if bytes > 0 {
            must(unsafe { ec_alloc_async(&mut p, bytes, stream()) }, "ec_alloc_async");
        }
        Self { ptr: p }
    }
    pub(crate) fn ptr(&self) -> *mut c_void { self.ptr }
}
impl Drop for DeviceMem {
    fn drop(&mut self) {
        if !self.ptr.is_null() {
            unsafe { ec_free_async(self.ptr, stream()) };
        }
    }
}
