//! `MaskedCellBuffer` and `NoData` (src/masked/masked_buffer.rs, src/masked/nodata.rs).
use crate::ffi::*;
use crate::{check, must, stream, CellBuffer, CellEncoding, CellType, CellValue, Mask, Result};
use std::ops::{Add, Div, Mul, Neg, Sub};
use std::os::raw::c_void;

/// src/masked/nodata.rs:9-21
#[derive(Debug, Copy, Clone)]
pub enum NoData<T: CellEncoding> {
    /// the cell type's conventional nodata value (minimum for signed integers and floats' NaN, …)
    Default,
    Value(T),
    /// nothing is nodata
    None,
}

impl<T: CellEncoding> NoData<T> {
    pub fn new(value: T) -> Self { NoData::Value(value) }
    /// src/masked/nodata.rs:23-40
    pub fn value(&self) -> Option<CellValue> {
        match self {
            NoData::None => None,
            NoData::Value(v) => Some((*v).into()),
            NoData::Default => {
                let mut d = CellValue::UInt8(0).to_ffi();
                must(unsafe { ec_nodata_default(T::cell_type() as u8, &mut d) }, "ec_nodata_default");
                Some(CellValue::from_ffi(&d))
            }
        }
    }
    /// src/masked/nodata.rs:42-49 (bitwise equality: a NaN nodata matches NaN cells)
    pub fn is(&self, value: CellValue) -> bool { self.value().map_or(false, |nd| nd == value) }
}

pub struct MaskedCellBuffer {
    buffer: CellBuffer,
    mask: Mask,
}

impl MaskedCellBuffer {
    /// src/masked/masked_buffer.rs:40-47 — the lengths must agree.
    pub fn new(buffer: CellBuffer, mask: Mask) -> Self {
        assert_eq!(buffer.len(), mask.len(), "Mask and buffer must have the same length");
        Self { buffer, mask }
    }
    /// src/masked/masked_buffer.rs:62-71 — the mask is computed on the device from the uploaded cells.
    pub fn from_vec_with_nodata<T: CellEncoding>(data: Vec<T>, nodata: NoData<T>) -> Self {
        let buffer = CellBuffer::from_vec(data);
        let mask = Mask::uninit(buffer.len());
        let nd = nodata.value().map(|v| v.to_ffi());
        let nd_ptr = nd.as_ref().map_or(std::ptr::null(), |v| v as *const ec_value);
        must(unsafe { ec_mask_from_nodata(buffer.ct as u8, buffer.dev_ptr(), buffer.len(), nd_ptr, mask.dev_ptr_mut(), stream()) },
             "ec_mask_from_nodata");
        Self { buffer, mask }
    }
    pub fn fill(len: usize, value: CellValue) -> Self { CellBuffer::fill(len, value).into() }
    pub fn with_defaults(len: usize, ct: CellType) -> Self { CellBuffer::with_defaults(len, ct).into() }

    pub fn buffer(&self) -> &CellBuffer { &self.buffer }
    pub fn mask(&self) -> &Mask { &self.mask }
    pub fn len(&self) -> usize { self.buffer.len() }
    pub fn is_empty(&self) -> bool { self.buffer.is_empty() }
    pub fn cell_type(&self) -> CellType { self.buffer.cell_type() }
    pub fn get(&self, index: usize) -> CellValue { self.buffer.get(index) }
    /// src/masked/masked_buffer.rs:100-106
    pub fn get_masked(&self, index: usize) -> Option<CellValue> {
        if self.mask.get(index) { Some(self.buffer.get(index)) } else { None }
    }
    /// src/masked/masked_buffer.rs:132-134
    pub fn counts(&self) -> (usize, usize) { self.mask.counts() }
    /// src/masked/masked_buffer.rs:200-206
    pub fn convert(&self, cell_type: CellType) -> Result<Self> {
        Ok(Self { buffer: self.buffer.convert(cell_type)?, mask: self.mask.clone() })
    }
    /// src/masked/masked_buffer.rs:208-217 — masked-out cells do not take part.
    pub fn min_max(&self) -> (CellValue, CellValue) {
        let (mut mn, mut mx) = (CellValue::UInt8(0).to_ffi(), CellValue::UInt8(0).to_ffi());
        must(unsafe { ec_min_max(self.buffer.ct as u8, self.buffer.dev_ptr(), self.mask.dev_ptr(), self.len(), &mut mn, &mut mx, stream()) },
             "ec_min_max");
        (CellValue::from_ffi(&mn), CellValue::from_ffi(&mx))
    }
    pub fn to_vec<T: CellEncoding>(&self) -> Result<Vec<T>> { self.buffer.to_vec() }
    /// src/masked/masked_buffer.rs:137-152 — masked-out cells come back as the nodata value.
    pub fn to_vec_with_nodata<T: CellEncoding>(&self, no_data: NoData<T>) -> Result<Vec<T>> {
        let conv = self.buffer.convert(T::cell_type())?;
        assert_eq!(conv.cell_type(), T::cell_type());
        let nd = match no_data.value() {
            None => return conv.to_vec(),
            Some(v) => v.to_ffi(),
        };
        let sel = CellBuffer::uninit(conv.cell_type(), conv.len());
        check(unsafe { ec_mask_select(conv.ct as u8, conv.dev_ptr(), self.mask.dev_ptr(), conv.len(), &nd, sel.mem.ptr(), stream()) })?;
        let mut v = Vec::<T>::with_capacity(sel.len());
        if sel.len() > 0 {
            check(unsafe { ec_download(v.as_mut_ptr() as *mut c_void, sel.dev_ptr(), sel.len() * std::mem::size_of::<T>(), stream()) })?;
        }
        unsafe { v.set_len(sel.len()) };
        Ok(v)
    }

    fn binop(&self, op: ec_op, rhs: &Self) -> Self {
        let n = self.len().min(rhs.len());
        if n == 0 {
            return Self { buffer: CellBuffer::empty_u8(), mask: Mask::uninit(0) };
        }
        // value op and `lmask & rmask` in one launch (src/masked/masked_buffer.rs:326-335)
        let (out, om) = (CellBuffer::uninit(CellType::Float64, n), Mask::uninit(n));
        must(unsafe {
            ec_masked_binop(op, self.buffer.ct as u8, self.buffer.dev_ptr(), self.mask.dev_ptr(), rhs.buffer.ct as u8,
                            rhs.buffer.dev_ptr(), rhs.mask.dev_ptr(), n, out.mem.ptr() as *mut f64, om.dev_ptr_mut(), stream())
        }, "ec_masked_binop");
        Self { buffer: out, mask: om }
    }
}

/// `From<CellBuffer>` (src/masked/masked_buffer.rs:250-255): every cell valid.
impl From<CellBuffer> for MaskedCellBuffer {
    fn from(buffer: CellBuffer) -> Self {
        let mask = Mask::fill(buffer.len(), true);
        Self { buffer, mask }
    }
}

impl Clone for MaskedCellBuffer {
    fn clone(&self) -> Self { Self { buffer: self.buffer.clone(), mask: self.mask.clone() } }
}

macro_rules! mcb_bin_op {
    ($trt:ident, $mth:ident, $op:expr) => {
        impl $trt for &MaskedCellBuffer {
            type Output = MaskedCellBuffer;
            fn $mth(self, rhs: Self) -> MaskedCellBuffer { self.binop($op, rhs) }
        }
        impl $trt for MaskedCellBuffer {
            type Output = MaskedCellBuffer;
            fn $mth(self, rhs: Self) -> MaskedCellBuffer { (&self).binop($op, &rhs) }
        }
        // RHS scalar: the mask is carried over unchanged
        impl<R: Into<CellValue>> $trt<R> for MaskedCellBuffer {
            type Output = MaskedCellBuffer;
            fn $mth(self, rhs: R) -> MaskedCellBuffer {
                MaskedCellBuffer { buffer: self.buffer.binop_scalar($op, rhs.into()), mask: self.mask }
            }
        }
    };
}
mcb_bin_op!(Add, add, EC_ADD);
mcb_bin_op!(Sub, sub, EC_SUB);
mcb_bin_op!(Mul, mul, EC_MUL);
mcb_bin_op!(Div, div, EC_DIV);

impl Neg for &MaskedCellBuffer {
    type Output = MaskedCellBuffer;
    fn neg(self) -> MaskedCellBuffer { MaskedCellBuffer { buffer: -&self.buffer, mask: self.mask.clone() } }
}
