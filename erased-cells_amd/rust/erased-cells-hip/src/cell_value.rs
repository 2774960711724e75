//! `CellValue` (src/value.rs:12-20): a tagged scalar.  Scalar arithmetic stays on the host as in the
//! reference; this file only adds the 16-byte `ec_value` image used across the ABI.
use crate::ffi::{ec_value, ec_value_convert, ec_value_to_f64};
use crate::{check, CellType, Result};

#[derive(Debug, Copy, Clone)]
pub enum CellValue {
    UInt8(u8), UInt16(u16), UInt32(u32), UInt64(u64), Int8(i8), Int16(i16), Int32(i32), Int64(i64),
    Float32(f32), Float64(f64),
}

macro_rules! from_primitive {
    ($(($id:ident, $p:ident)),*) => { $(
        impl From<$p> for CellValue { fn from(v: $p) -> Self { CellValue::$id(v) } }
    )* }
}
from_primitive!((UInt8, u8), (UInt16, u16), (UInt32, u32), (UInt64, u64), (Int8, i8), (Int16, i16), (Int32, i32),
                (Int64, i64), (Float32, f32), (Float64, f64));

impl CellValue {
    pub fn new<T: Into<CellValue>>(v: T) -> Self { v.into() }

    pub fn cell_type(&self) -> CellType {
        match self {
            CellValue::UInt8(_) => CellType::UInt8, CellValue::UInt16(_) => CellType::UInt16,
            CellValue::UInt32(_) => CellType::UInt32, CellValue::UInt64(_) => CellType::UInt64,
            CellValue::Int8(_) => CellType::Int8, CellValue::Int16(_) => CellType::Int16,
            CellValue::Int32(_) => CellType::Int32, CellValue::Int64(_) => CellType::Int64,
            CellValue::Float32(_) => CellType::Float32, CellValue::Float64(_) => CellType::Float64,
        }
    }

    /// The C union is 8 little-endian bytes; narrower payloads occupy the low bytes.
    pub(crate) fn to_ffi(&self) -> ec_value {
        let bits: u64 = match *self {
            CellValue::UInt8(v) => v as u64, CellValue::UInt16(v) => v as u64, CellValue::UInt32(v) => v as u64,
            CellValue::UInt64(v) => v, CellValue::Int8(v) => v as u8 as u64, CellValue::Int16(v) => v as u16 as u64,
            CellValue::Int32(v) => v as u32 as u64, CellValue::Int64(v) => v as u64,
            CellValue::Float32(v) => v.to_bits() as u64, CellValue::Float64(v) => v.to_bits(),
        };
        ec_value { dtype: self.cell_type() as u8, pad_: [0; 7], bits }
    }

    pub(crate) fn from_ffi(v: &ec_value) -> Self {
        let b = v.bits;
        match CellType::from_code(v.dtype) {
            CellType::UInt8 => CellValue::UInt8(b as u8), CellType::UInt16 => CellValue::UInt16(b as u16),
            CellType::UInt32 => CellValue::UInt32(b as u32), CellType::UInt64 => CellValue::UInt64(b),
            CellType::Int8 => CellValue::Int8(b as u8 as i8), CellType::Int16 => CellValue::Int16(b as u16 as i16),
            CellType::Int32 => CellValue::Int32(b as u32 as i32), CellType::Int64 => CellValue::Int64(b as i64),
            CellType::Float32 => CellValue::Float32(f32::from_bits(b as u32)),
            CellType::Float64 => CellValue::Float64(f64::from_bits(b)),
        }
    }

    /// src/value.rs:74-98 — refused with `NarrowingError` unless the lattice allows it.
    pub fn convert(&self, cell_type: CellType) -> Result<CellValue> {
        let (src, mut dst) = (self.to_ffi(), CellValue::UInt8(0).to_ffi());
        check(unsafe { ec_value_convert(&src, cell_type as u8, &mut dst) })?;
        Ok(CellValue::from_ffi(&dst))
    }

    pub fn to_f64(&self) -> f64 {
        let v = self.to_ffi();
        unsafe { ec_value_to_f64(&v) }
    }
}

/// Equality as the reference defines it for cells of one type: bitwise (so a NaN nodata matches itself).
impl PartialEq for CellValue {
    fn eq(&self, other: &Self) -> bool {
        let (a, b) = (self.to_ffi(), other.to_ffi());
        a.dtype == b.dtype && a.bits == b.bits
    }
}
