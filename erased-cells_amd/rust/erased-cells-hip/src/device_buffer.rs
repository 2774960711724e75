//! `CellBuffer` with its cells resident in HBM (src/buffer.rs).
use crate::ffi::*;
use crate::{check, must, stream, CellEncoding, CellType, CellValue, DeviceMem, Error, Result};
use std::cmp::Ordering;
use std::ops::{Add, Div, Mul, Neg, Sub};
use std::os::raw::c_void;

pub struct CellBuffer {
    pub(crate) ct: CellType,
    pub(crate) len: usize,
    pub(crate) mem: DeviceMem,
}

impl CellBuffer {
    pub(crate) fn uninit(ct: CellType, len: usize) -> Self { Self { ct, len, mem: DeviceMem::new(len * ct.size_of()) } }
    /// What `collect()` of nothing yields in the reference (src/buffer.rs:233-234).
    pub(crate) fn empty_u8() -> Self { Self::uninit(CellType::UInt8, 0) }
    pub(crate) fn dev_ptr(&self) -> *const c_void { self.mem.ptr() }

    /// `From<Vec<T>>`: one host-to-HBM copy.
    pub fn from_vec<T: CellEncoding>(data: Vec<T>) -> Self {
        let out = Self::uninit(T::cell_type(), data.len());
        if !data.is_empty() {
            let bytes = data.len() * std::mem::size_of::<T>();
            must(unsafe { ec_upload(out.mem.ptr(), data.as_ptr() as *const c_void, bytes, stream()) }, "ec_upload");
        }
        out
    }
    /// src/buffer.rs:79-88
    pub fn fill(len: usize, value: CellValue) -> Self {
        let out = Self::uninit(value.cell_type(), len);
        let v = value.to_ffi();
        must(unsafe { ec_fill(out.ct as u8, out.mem.ptr(), len, &v, stream()) }, "ec_fill");
        out
    }
    /// src/buffer.rs:68-77
    pub fn with_defaults(len: usize, ct: CellType) -> Self {
        // T::default() per cell type (a u8 zero cannot be converted: UInt8 does not fit Int8)
        let zero = match ct {
            CellType::UInt8 => CellValue::UInt8(0), CellType::UInt16 => CellValue::UInt16(0),
            CellType::UInt32 => CellValue::UInt32(0), CellType::UInt64 => CellValue::UInt64(0),
            CellType::Int8 => CellValue::Int8(0), CellType::Int16 => CellValue::Int16(0),
            CellType::Int32 => CellValue::Int32(0), CellType::Int64 => CellValue::Int64(0),
            CellType::Float32 => CellValue::Float32(0.0), CellType::Float64 => CellValue::Float64(0.0),
        };
        Self::fill(len, zero)
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    pub fn cell_type(&self) -> CellType { self.ct }

    /// src/buffer.rs:125-134 — panics when `index` is out of bounds, as `Vec` indexing does.
    pub fn get(&self, index: usize) -> CellValue {
        assert!(index < self.len, "index out of bounds: the len is {} but the index is {}", self.len, index);
        let mut v = CellValue::UInt8(0).to_ffi();
        v.dtype = self.ct as u8;
        let sz = self.ct.size_of();
        let src = unsafe { (self.mem.ptr() as *const u8).add(index * sz) } as *const c_void;
        must(unsafe { ec_download(&mut v.bits as *mut u64 as *mut c_void, src, sz, stream()) }, "ec_download");
        CellValue::from_ffi(&v)
    }

    /// `BufferOps::convert` (src/buffer.rs:150-167)
    pub fn convert(&self, cell_type: CellType) -> Result<Self> {
        if !self.ct.can_fit_into(cell_type) {
            return Err(Error::NarrowingError { src: self.ct, dst: cell_type });
        }
        if cell_type != self.ct && self.len == 0 {
            return Ok(Self::empty_u8());
        }
        let out = Self::uninit(cell_type, self.len);
        check(unsafe { ec_convert(self.ct as u8, self.dev_ptr(), cell_type as u8, out.mem.ptr(), self.len, stream()) })?;
        Ok(out)
    }

    /// `BufferOps::min_max` (src/buffer.rs:169-173): total order, folded from (T::MAX, T::MIN).
    pub fn min_max(&self) -> (CellValue, CellValue) {
        let (mut mn, mut mx) = (CellValue::UInt8(0).to_ffi(), CellValue::UInt8(0).to_ffi());
        must(unsafe { ec_min_max(self.ct as u8, self.dev_ptr(), std::ptr::null(), self.len, &mut mn, &mut mx, stream()) }, "ec_min_max");
        (CellValue::from_ffi(&mn), CellValue::from_ffi(&mx))
    }

    /// `BufferOps::to_vec` (src/buffer.rs:175-185)
    pub fn to_vec<T: CellEncoding>(&self) -> Result<Vec<T>> {
        let r = self.convert(T::cell_type())?;
        assert_eq!(r.ct, T::cell_type());
        let mut v = Vec::<T>::with_capacity(r.len);
        if r.len > 0 {
            check(unsafe { ec_download(v.as_mut_ptr() as *mut c_void, r.dev_ptr(), r.len * std::mem::size_of::<T>(), stream()) })?;
        }
        unsafe { v.set_len(r.len) };
        Ok(v)
    }

    pub(crate) fn binop(&self, op: ec_op, rhs: &Self) -> Self {
        let n = self.len.min(rhs.len); // zip (src/buffer.rs:327)
        if n == 0 {
            return Self::empty_u8();
        }
        let out = Self::uninit(CellType::Float64, n); // every binop widens to f64 (src/value.rs:207)
        must(unsafe { ec_binop(op, self.ct as u8, self.dev_ptr(), rhs.ct as u8, rhs.dev_ptr(), n, out.mem.ptr() as *mut f64, stream()) },
             "ec_binop");
        out
    }
    pub(crate) fn binop_scalar(&self, op: ec_op, rhs: CellValue) -> Self {
        if self.len == 0 {
            return Self::empty_u8();
        }
        let out = Self::uninit(CellType::Float64, self.len);
        let v = rhs.to_ffi();
        must(unsafe { ec_binop_scalar(op, self.ct as u8, self.dev_ptr(), self.len, &v, out.mem.ptr() as *mut f64, stream()) },
             "ec_binop_scalar");
        out
    }
}

impl Clone for CellBuffer {
    fn clone(&self) -> Self {
        let out = Self::uninit(self.ct, self.len);
        if self.len > 0 {
            must(unsafe { ec_copy(out.mem.ptr(), self.dev_ptr(), self.len * self.ct.size_of(), stream()) }, "ec_copy");
        }
        out
    }
}

impl<T: CellEncoding> From<Vec<T>> for CellBuffer {
    fn from(v: Vec<T>) -> Self { CellBuffer::from_vec(v) }
}

// cb_bin_op! (src/buffer.rs:321-358): the iterator-chain bodies become one FFI call.
macro_rules! cb_bin_op {
    ($trt:ident, $mth:ident, $op:expr) => {
        impl $trt for &CellBuffer {
            type Output = CellBuffer;
            fn $mth(self, rhs: Self) -> CellBuffer { self.binop($op, rhs) }
        }
        impl $trt for CellBuffer {
            type Output = CellBuffer;
            fn $mth(self, rhs: Self) -> CellBuffer { (&self).binop($op, &rhs) }
        }
        // RHS borrow (src/buffer.rs:338-343)
        impl $trt<&CellBuffer> for CellBuffer {
            type Output = CellBuffer;
            fn $mth(self, rhs: &CellBuffer) -> CellBuffer { (&self).binop($op, rhs) }
        }
        // RHS scalar (src/buffer.rs:346-352)
        impl<R: Into<CellValue>> $trt<R> for CellBuffer {
            type Output = CellBuffer;
            fn $mth(self, rhs: R) -> CellBuffer { self.binop_scalar($op, rhs.into()) }
        }
    };
}
cb_bin_op!(Add, add, EC_ADD);
cb_bin_op!(Sub, sub, EC_SUB);
cb_bin_op!(Mul, mul, EC_MUL);
cb_bin_op!(Div, div, EC_DIV);

impl Neg for &CellBuffer {
    type Output = CellBuffer;
    /// src/buffer.rs:360-365; the result variant widens per src/value.rs:224-240
    fn neg(self) -> CellBuffer {
        if self.len == 0 {
            return CellBuffer::empty_u8();
        }
        let out = CellBuffer::uninit(CellType::from_code(unsafe { ec_neg_result_type(self.ct as u8) }), self.len);
        must(unsafe { ec_neg(self.ct as u8, self.dev_ptr(), self.len, out.mem.ptr(), stream()) }, "ec_neg");
        out
    }
}

// impl Ord / PartialEq for CellBuffer (src/buffer.rs:373-436): first differing cell found on the device.
impl PartialEq for CellBuffer {
    fn eq(&self, other: &Self) -> bool { self.cmp(other) == Ordering::Equal }
}
impl Eq for CellBuffer {}
impl PartialOrd for CellBuffer {
    fn partial_cmp(&self, other: &Self) -> Option<Ordering> { Some(self.cmp(other)) }
}
impl Ord for CellBuffer {
    fn cmp(&self, other: &Self) -> Ordering {
        let mut o = 0i32;
        must(unsafe { ec_buffer_cmp(self.ct as u8, self.dev_ptr(), self.len, other.ct as u8, other.dev_ptr(), other.len, &mut o, stream()) },
             "ec_buffer_cmp");
        o.cmp(&0)
    }
}
