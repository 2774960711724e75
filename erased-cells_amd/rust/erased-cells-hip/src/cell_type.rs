//! `CellType` (src/ctype.rs:11-180 of the reference): the ten cell encodings.  `self as u8` IS the ABI dtype
//! code; the lattice (`union`, `can_fit_into`, limits) is answered by the library's host-side table, the same
//! one its kernel dispatch uses.
use crate::error::Error;
use crate::ffi::*;
use crate::CellValue;
use std::fmt::{Debug, Display, Formatter};
use std::str::FromStr;

/// Cell-type variants, in the order of `with_ct!`.
#[derive(Debug, Copy, Clone, PartialEq, Eq, PartialOrd, Ord, Hash)]
#[repr(u8)]
pub enum CellType {
    UInt8 = 0,
    UInt16 = 1,
    UInt32 = 2,
    UInt64 = 3,
    Int8 = 4,
    Int16 = 5,
    Int32 = 6,
    Int64 = 7,
    Float32 = 8,
    Float64 = 9,
}

const ALL: [CellType; 10] = [
    CellType::UInt8, CellType::UInt16, CellType::UInt32, CellType::UInt64, CellType::Int8, CellType::Int16,
    CellType::Int32, CellType::Int64, CellType::Float32, CellType::Float64,
];

/// `Display` is the same as `Debug`.
impl Display for CellType {
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        Debug::fmt(self, f)
    }
}

impl FromStr for CellType {
    type Err = Error;

    fn from_str(s: &str) -> Result<Self, Self::Err> {
        ALL.iter().copied().find(|ct| ct.to_string() == s).ok_or_else(|| Error::ParseError(s.into(), "CellType"))
    }
}

impl CellType {
    pub(crate) fn from_code(c: u8) -> Self {
        assert!((c as usize) < ALL.len(), "dtype code {c} out of range");
        ALL[c as usize]
    }

    /// Get an iterator over all the valid enumeration values.
    pub fn iter() -> impl Iterator<Item = CellType> {
        ALL.into_iter()
    }

    /// Determine if `self` is integral or floating-point.
    pub fn is_integral(&self) -> bool {
        !matches!(self, CellType::Float32 | CellType::Float64)
    }

    /// Determine if `self` is signed or unsigned.
    pub fn is_signed(&self) -> bool {
        !matches!(self, CellType::UInt8 | CellType::UInt16 | CellType::UInt32 | CellType::UInt64)
    }

    /// Number of bytes needed to encode `self`.
    pub fn size_of(&self) -> usize {
        unsafe { ec_size_of(*self as u8) }
    }

    /// Select the `CellType` that can numerically contain both `self` and `other`.
    pub fn union(self, other: Self) -> Self {
        Self::from_code(unsafe { ec_union(self as u8, other as u8) })
    }

    /// Determine of `self` can fit within `other`.
    pub fn can_fit_into(self, other: Self) -> bool {
        unsafe { ec_can_fit_into(self as u8, other as u8) != 0 }
    }

    /// Construct the zero value for a variant.
    pub fn zero(&self) -> CellValue {
        CellValue::small(*self, 0)
    }

    /// Construct the one value for a variant.
    pub fn one(&self) -> CellValue {
        CellValue::small(*self, 1)
    }

    /// Determine the minimum value that can be represented by `self`.
    pub fn min_value(&self) -> CellValue {
        let mut v = CellValue::UInt8(0).to_ffi();
        unsafe { ec_min_value(*self as u8, &mut v) };
        CellValue::from_ffi(&v)
    }

    /// Determine the maximum value that can be represented by `self`.
    pub fn max_value(&self) -> CellValue {
        let mut v = CellValue::UInt8(0).to_ffi();
        unsafe { ec_max_value(*self as u8, &mut v) };
        CellValue::from_ffi(&v)
    }
}
