//! [`CellType`]: the ten cell encodings.
//!
//! PROVENANCE.  The enum (variant names and order) and the method signatures are the reference's public surface
//! (erased-cells 0.1.1, src/ctype.rs:11-180, MIT License, Copyright (c) 2023 Astraea, Inc.); the bodies are this crate's:
//! `self as u8` IS the ABI dtype code, and the lattice (`union`, `can_fit_into`, sizes, limits) is answered by the
//! library's host-side table — the same one its kernel dispatch uses — instead of being restated here.
use crate::error::Error;
use crate::ffi::*;
use crate::CellValue;
use std::fmt::{Debug, Display, Formatter};
use std::str::FromStr;

// api-surface(src/ctype.rs:9-20): the enum — variant names and order (discriminants written out: they are the ABI's dtype codes)
/// The encoding of a cell: one of the ten Rust primitives a cell can be.
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
// end api-surface

const ALL: [CellType; 10] = [
    CellType::UInt8, CellType::UInt16, CellType::UInt32, CellType::UInt64, CellType::Int8, CellType::Int16,
    CellType::Int32, CellType::Int64, CellType::Float32, CellType::Float64,
];

/// Prints the variant name, as `Debug` does.
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

    /// The ten variants in discriminant order.
    pub fn iter() -> impl Iterator<Item = CellType> {
        ALL.into_iter()
    }

    /// `false` for the two float types.
    pub fn is_integral(&self) -> bool {
        !matches!(self, CellType::Float32 | CellType::Float64)
    }

    /// `false` for the four unsigned integer types (floats are signed).
    pub fn is_signed(&self) -> bool {
        !matches!(self, CellType::UInt8 | CellType::UInt16 | CellType::UInt32 | CellType::UInt64)
    }

    /// Bytes per cell.
    pub fn size_of(&self) -> usize {
        unsafe { ec_size_of(*self as u8) }
    }

    /// The narrowest cell type that represents every value of both `self` and `other` (where none does — e.g. `u64` with
    /// a signed type — `Float64`).
    pub fn union(self, other: Self) -> Self {
        Self::from_code(unsafe { ec_union(self as u8, other as u8) })
    }

    /// Does `other` represent every value of `self`?  (`self.union(other) == other`.)  This is what `convert` checks.
    pub fn can_fit_into(self, other: Self) -> bool {
        unsafe { ec_can_fit_into(self as u8, other as u8) != 0 }
    }

    /// 0 as a value of this cell type.
    pub fn zero(&self) -> CellValue {
        CellValue::small(*self, 0)
    }

    /// 1 as a value of this cell type.
    pub fn one(&self) -> CellValue {
        CellValue::small(*self, 1)
    }

    /// The least value of the primitive (`T::MIN`: finite for floats).
    pub fn min_value(&self) -> CellValue {
        let mut v = CellValue::UInt8(0).to_ffi();
        unsafe { ec_min_value(*self as u8, &mut v) };
        CellValue::from_ffi(&v)
    }

    /// The greatest value of the primitive (`T::MAX`: finite for floats).
    pub fn max_value(&self) -> CellValue {
        let mut v = CellValue::UInt8(0).to_ffi();
        unsafe { ec_max_value(*self as u8, &mut v) };
        CellValue::from_ffi(&v)
    }
}
