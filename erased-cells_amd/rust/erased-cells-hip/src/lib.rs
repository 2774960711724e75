//! Device-resident `CellBuffer` whose operator bodies are calls into liberased_cells_hip.so.
//!
//! The shape follows erased-cells 0.1.1: `CellType` keeps its discriminants (they ARE the ABI dtype
//! codes), the host keeps the type tag, zip truncation and the empty-result-is-UInt8 rule, and each
//! `impl Add/Sub/Mul/Div for &CellBuffer` body becomes one `ec_binop` call.
//! NOT COMPILED in the build image (no rustc); see INTEGRATION.md.
pub mod ffi;

use ffi::*;
use std::ffi::CStr;
use std::ops::{Add, Div, Mul, Neg, Sub};
use std::os::raw::c_void;
use std::ptr;

#[derive(Debug, Copy, Clone, PartialEq, Eq, PartialOrd, Ord)]
#[repr(u8)]
pub enum CellType { UInt8, UInt16, UInt32, UInt64, Int8, Int16, Int32, Int64, Float32, Float64 }

impl CellType {
    pub fn size_of(self) -> usize { [1, 2, 4, 8, 1, 2, 4, 8, 4, 8][self as usize] }
    fn from_code(c: u8) -> Self { unsafe { std::mem::transmute(c) } }
    pub fn union(self, other: Self) -> Self { Self::from_code(unsafe { ec_union(self as u8, other as u8) }) }
    pub fn can_fit_into(self, other: Self) -> bool { unsafe { ec_can_fit_into(self as u8, other as u8) != 0 } }
}

#[derive(thiserror::Error, Debug)]
pub enum Error {
    #[error("Invalid narrowing from cell-type {src:?} to {dst:?}")]
    NarrowingError { src: CellType, dst: CellType },
    #[error("HIP backend: {0}")]
    Backend(String),
}
pub type Result<T> = std::result::Result<T, Error>;

fn check(st: ec_status) -> Result<()> {
    if st == EC_OK { return Ok(()); }
    if st == EC_ERR_NARROWING {
        let (mut s, mut d) = (0u8, 0u8);
        unsafe { ec_last_narrowing(&mut s, &mut d) };
        return Err(Error::NarrowingError { src: CellType::from_code(s), dst: CellType::from_code(d) });
    }
    Err(Error::Backend(unsafe { CStr::from_ptr(ec_last_error_string()) }.to_string_lossy().into_owned()))
}

/// One HBM allocation from the stream-ordered pool (operator results are allocated per call, as the
/// reference `collect()`s a fresh Vec; hipMalloc/hipFree per operator would cost as much as the kernel).
struct DeviceMem(*mut c_void);
impl DeviceMem {
    fn new(bytes: usize) -> Result<Self> { let mut p = ptr::null_mut(); check(unsafe { ec_alloc_async(&mut p, bytes, ptr::null_mut()) })?; Ok(Self(p)) }
}
impl Drop for DeviceMem { fn drop(&mut self) { unsafe { ec_free_async(self.0, ptr::null_mut()) }; } }

/// `CellBuffer` with its cells resident on the GPU.
pub struct CellBuffer { ct: CellType, len: usize, mem: DeviceMem }

pub trait CellEncoding: Copy { fn cell_type() -> CellType; }
macro_rules! encoding { ($(($id:ident, $p:ident)),*) => { $(impl CellEncoding for $p { fn cell_type() -> CellType { CellType::$id } })* } }
encoding!((UInt8, u8), (UInt16, u16), (UInt32, u32), (UInt64, u64), (Int8, i8), (Int16, i16), (Int32, i32), (Int64, i64), (Float32, f32), (Float64, f64));

impl CellBuffer {
    pub fn from_vec<T: CellEncoding>(data: Vec<T>) -> Result<Self> {
        let bytes = data.len() * std::mem::size_of::<T>();
        let mem = DeviceMem::new(bytes)?;
        check(unsafe { ec_upload(mem.0, data.as_ptr() as *const c_void, bytes, ptr::null_mut()) })?;
        Ok(Self { ct: T::cell_type(), len: data.len(), mem })
    }
    fn empty(ct: CellType, len: usize) -> Result<Self> { Ok(Self { ct, len, mem: DeviceMem::new(len * ct.size_of())? }) }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    pub fn cell_type(&self) -> CellType { self.ct }

    /// BufferOps::convert (src/buffer.rs:150-167)
    pub fn convert(&self, cell_type: CellType) -> Result<Self> {
        if !self.ct.can_fit_into(cell_type) { return Err(Error::NarrowingError { src: self.ct, dst: cell_type }); }
        if cell_type != self.ct && self.len == 0 { return Self::empty(CellType::UInt8, 0); } // buffer.rs:233-234
        let out = Self::empty(cell_type, self.len)?;
        check(unsafe { ec_convert(self.ct as u8, self.mem.0, cell_type as u8, out.mem.0, self.len, ptr::null_mut()) })?;
        Ok(out)
    }
    /// BufferOps::to_vec (src/buffer.rs:175-185)
    pub fn to_vec<T: CellEncoding>(&self) -> Result<Vec<T>> {
        let r = self.convert(T::cell_type())?;
        assert_eq!(r.ct, T::cell_type());
        let mut v = Vec::<T>::with_capacity(r.len);
        check(unsafe { ec_download(v.as_mut_ptr() as *mut c_void, r.mem.0, r.len * std::mem::size_of::<T>(), ptr::null_mut()) })?;
        unsafe { v.set_len(r.len) };
        Ok(v)
    }
    fn binop(&self, op: ec_op, rhs: &Self) -> Self {
        let n = self.len.min(rhs.len); // zip (src/buffer.rs:327)
        if n == 0 { return Self::empty(CellType::UInt8, 0).expect("alloc"); }
        let out = Self::empty(CellType::Float64, n).expect("alloc");
        check(unsafe { ec_binop(op, self.ct as u8, self.mem.0, rhs.ct as u8, rhs.mem.0, n, out.mem.0 as *mut f64, ptr::null_mut()) })
            .expect("ec_binop"); // arithmetic is infallible in the reference; a backend failure is a panic
        out
    }
}

// cb_bin_op! (src/buffer.rs:321-358): the iterator-chain bodies become one FFI call.
macro_rules! cb_bin_op { ($trt:ident, $mth:ident, $op:expr) => {
    impl $trt for &CellBuffer { type Output = CellBuffer; fn $mth(self, rhs: Self) -> CellBuffer { self.binop($op, rhs) } }
    impl $trt for CellBuffer { type Output = CellBuffer; fn $mth(self, rhs: Self) -> CellBuffer { (&self).binop($op, &rhs) } }
    impl $trt<&CellBuffer> for CellBuffer { type Output = CellBuffer; fn $mth(self, rhs: &CellBuffer) -> CellBuffer { (&self).binop($op, rhs) } }
} }
cb_bin_op!(Add, add, EC_ADD);
cb_bin_op!(Sub, sub, EC_SUB);
cb_bin_op!(Mul, mul, EC_MUL);
cb_bin_op!(Div, div, EC_DIV);

// impl Ord / PartialEq for CellBuffer (src/buffer.rs:373-436): decided on the device.
impl PartialEq for CellBuffer {
    fn eq(&self, other: &Self) -> bool { self.cmp(other) == std::cmp::Ordering::Equal }
}
impl Eq for CellBuffer {}
impl PartialOrd for CellBuffer {
    fn partial_cmp(&self, other: &Self) -> Option<std::cmp::Ordering> { Some(self.cmp(other)) }
}
impl Ord for CellBuffer {
    fn cmp(&self, other: &Self) -> std::cmp::Ordering {
        let mut o = 0i32;
        check(unsafe { ec_buffer_cmp(self.ct as u8, self.mem.0, self.len, other.ct as u8, other.mem.0, other.len, &mut o, ptr::null_mut()) })
            .expect("ec_buffer_cmp");
        o.cmp(&0)
    }
}

impl Neg for &CellBuffer {
    type Output = CellBuffer;
    fn neg(self) -> CellBuffer { // src/buffer.rs:360-365
        if self.len == 0 { return CellBuffer::empty(CellType::UInt8, 0).expect("alloc"); }
        let out = CellBuffer::empty(CellType::from_code(unsafe { ec_neg_result_type(self.ct as u8) }), self.len).expect("alloc");
        check(unsafe { ec_neg(self.ct as u8, self.mem.0, self.len, out.mem.0, ptr::null_mut()) }).expect("ec_neg");
        out
    }
}
